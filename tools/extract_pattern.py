"""Dev-time tool (runs only in the build container): extracts the rBRIEF sampling-point DATA
table from the reference tree into include/sgx_orb_pattern.h as a flat int8 array.
Only the 1024 integers are taken; no reference code is copied."""
import re, sys
src = open('/root/reference/src/sg-slam/src/ORBextractor.cc').read()
i = src.index('static int bit_pattern_31_[256*4]')
body = src[src.index('{', i) + 1:src.index('};', i)]
body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
nums = [int(x) for x in re.findall(r'-?\d+', body)]
assert len(nums) == 1024
hdr = open('/root/repo/include/sgx_orb_pattern.h').read()
head = hdr[:hdr.index('= {') + 3]
lines = ['  ' + ','.join('%d' % v for v in nums[r:r + 32]) + ',' for r in range(0, 1024, 32)]
open('/root/repo/include/sgx_orb_pattern.h', 'w').write(head + '\n' + '\n'.join(lines) + '\n};\n#endif\n')
