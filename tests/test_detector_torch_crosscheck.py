"""Build-container cross-check of the ORACLE detector's layer semantics against an independent implementation (VERDICT r3 'next round' #8).

ncnn is absent (parity unpinned at the ncnn boundary), but PyTorch is in the image: the shipped graph
(tests/golden/mobilenetv3_ssdlite_voc.param) is executed a second time with torch.nn.functional.conv2d in float64 — torch's own padding, stride
and group handling, [outc][inc / group][k][k] weight layout, permute / flatten / cat / softmax — and every layer's output (on the oracle's own inputs of that layer) is compared with the oracle's float64 run
(oracle/detector_oracle.py::forward, which uses hand-written im2col + matmul).  Agreement to 1e-12 of the blob's magnitude means that what the oracle calls a
Convolution(0=outc 1=k 3=stride 4=pad 7=group), a ConvolutionDepthWise, a Permute(order 3), a Flatten / Concat of the SSD heads and a Softmax over the class axis
is what an independent framework computes for the same definitions; it does not pin ncnn's own arithmetic order (that needs tools/pin_third_party.py on a machine
with ncnn).  CPU only, ~10 s."""
import os
import numpy as np
import pytest
from oracle import detector_oracle as D

torch = pytest.importorskip('torch')
F = torch.nn.functional
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')


def torch_layer(L, W, get):
    """one graph layer with torch operators (float64) on the inputs `get(name)` returns; None for the non-tensor layers"""
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64))
    t, p, ins = L['type'], L['p'], L['ins']
    if t in ('Input', 'PriorBox', 'DetectionOutput', 'MemoryData', 'Split'): return None
    a = t64(get(ins[0]))
    if t in ('Convolution', 'ConvolutionDepthWise'):
        w, b = W[L['name']]
        outc, k, stride, pad, group = p[0], p[1], p.get(3, 1), p.get(4, 0), p.get(7, 1)
        return F.conv2d(a[None], t64(w).reshape(outc, a.shape[0] // group, k, k), t64(b), stride=stride, padding=pad, groups=group)[0]
    if t == 'BinaryOp':
        b = t64(get(ins[1])); op = p.get(0, 0)
        if b.numel() == 1: b = b.reshape(())
        return a + b if op == 0 else a * b if op == 2 else a / b
    if t == 'Clip': return torch.clamp(a, p[0], p[1])
    if t == 'ReLU': return torch.relu(a)
    if t == 'Permute': assert p[0] == 3; return a.permute(1, 2, 0).contiguous()                  # order 3: (c, h, w) -> (h, w, c)
    if t == 'Flatten': return a.reshape(-1)
    if t == 'Concat': return torch.cat([t64(get(i)).reshape(-1) if p.get(0, 0) == 0 else t64(get(i)) for i in ins], 0 if p.get(0, 0) == 0 else 1)
    if t == 'Reshape': return a.reshape(-1, p[0])
    if t == 'Softmax': return torch.softmax(a, 1)
    raise NotImplementedError(t)


@pytest.mark.parametrize('seed', [0, 5])
def test_oracle_layers_equal_torch_float64(seed):
    """layer by layer on the ORACLE's own input blobs (a whole-network comparison would measure the graph's drift amplification — 1e8 from the stem to the heads with
    these weights, visible even in float64 — not the layer definitions)"""
    layers = D.parse_param(PARAM)
    W, _ = D.synth_weights(layers, seed=7)
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)
    x = D.preprocess(img)
    _, ob = D.forward(layers, W, x, dt=np.float64)
    checked = 0; kinds = set()
    for L in layers:
        if L['name'] == 'mbox_priorbox': continue                  # concat of PriorBox outputs: host constants, not a tensor layer
        y = torch_layer(L, W, lambda n: ob[n])
        if y is None: continue
        a = np.asarray(ob[L['outs'][0]], np.float64).reshape(-1); b = y.numpy().reshape(-1)
        assert a.shape == b.shape, (L['name'], a.shape, b.shape)
        assert np.abs(a - b).max() <= 1e-12 * max(np.abs(a).max(), 1e-30), (L['type'], L['name'], float(np.abs(a - b).max()), float(np.abs(a).max()))
        checked += 1; kinds.add(L['type'])
    assert checked > 270 and {'Convolution', 'ConvolutionDepthWise', 'BinaryOp', 'Clip', 'ReLU', 'Permute', 'Flatten', 'Concat', 'Reshape', 'Softmax'} <= kinds, (checked, kinds)
