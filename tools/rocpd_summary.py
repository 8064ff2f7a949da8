"""Dump the per-kernel summary (top_kernels view) of a rocprofv3 rocpd .db into a text file for profiles/."""
import sqlite3
import sys


def main(db, out, note=''):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w') as f:
        if note:
            f.write(note + '\n')
        f.write('rocprofv3 --kernel-trace --stats  (durations in microseconds)\n')
        f.write(f'{"calls":>6} {"total_us":>12} {"avg_us":>10} {"pct":>6}  kernel\n')
        for name, calls, tot, avg, pct in rows:
            f.write(f'{calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {name[:150]}\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], ' '.join(sys.argv[3:]))
