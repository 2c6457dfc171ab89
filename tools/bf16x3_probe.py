"""bf16x3 probe, one shape (VERDICT r5 'Next' #3): ResNet-101 layer3 conv3, 256 -> 1024 on 8 frames of 30x54, on the trunk's REAL activation statistics.

  X = the input of resnet.layer3[0].conv3 for 8 synthetic 480x854 frames: the shipped trunk up to tap 'layer3', then that block's conv1 + bn1 + ReLU and
      conv2 (3x3, stride 2) + bn2 + ReLU through the shipped conv kernels;   W = that block's conv3 weight (seeded synthetic trunk, as bench.py).
  reference     fp64 product (torch, GPU)
  fp32 MFMA     the shipped kernel (ops.conv2d -> k_conv_igemm<64,64,2,4,1>, v_mfma_f32_16x16x4_f32: bitwise an fmaf chain per output)
  bf16 x3       tools/bf16x3_probe.hip: both operands as three bf16 pieces, 6 piece products on v_mfma_f32_32x32x16_bf16 (also 1, 3 and 9 products)

Prints rate (HIP events, 40 launches each, kernel alone) and max / rms error against fp64, relative to the rms of the output.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/bf16x3_probe.hip -o tools/_bin/libbf16x3.so;  python tools/bf16x3_probe.py
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frtm_vos_amd import ops  # noqa: E402
from frtm_vos_amd.lib.synthetic import SyntheticSequence  # noqa: E402
from frtm_vos_amd.model.feature_extractor import ResnetFeatureExtractor  # noqa: E402

DEV = 'cuda:0'
L = ctypes.CDLL(os.path.join(ROOT, 'tools', '_bin', 'libbf16x3.so'))
P = ctypes.c_void_p


def fold(bn):
    sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
    return sc.to(DEV).contiguous(), (bn.bias - bn.running_mean * sc).float().to(DEV).contiguous()


def conv_bn_relu(x, conv, bn, stride=1, pad=0):
    wT, ktab, lay = ops.pack_weights(conv.weight.data.to(DEV))
    sc, sh = fold(bn)
    return ops.conv2d(x, wT, conv.weight.shape[0], conv.weight.shape[2], stride, pad, ktab=ktab, scale=sc, shift=sh, relu=True, w_layout=lay)


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps          # us


def main():
    torch.set_grad_enabled(False)
    ext = ResnetFeatureExtractor('resnet101').to(DEV)
    seq = SyntheticSequence('probe', 8, (480, 854), 2, seed=5)
    frames = torch.stack([seq[t][0] for t in range(8)]).to(DEV)
    tap = ext(frames, ['layer3'])['layer3']                          # (8, 512, 60, 107)
    blk = ext.resnet.layer3[0]
    t1 = conv_bn_relu(tap, blk.conv1, blk.bn1)
    X = conv_bn_relu(t1, blk.conv2, blk.bn2, stride=2, pad=1).contiguous()      # (8, 256, 30, 54)
    W = blk.conv3.weight.data.reshape(1024, 256).to(DEV).contiguous()
    imgs, K, h, w = X.shape
    M, npix = W.shape[0], h * w
    N = imgs * npix
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    print('# tools/bf16x3_probe.py: C[%d][%d] = W[%d][%d] X, X = input of resnet101.layer3[0].conv3 on 8 synthetic 480x854 frames' % (M, N, M, K))
    print('# activations: mean %.3f  rms %.3f  max %.2f  zeros %.1f %%;  weights: rms %.4f  max |w| %.3f' %
          (float(X.mean()), float(X.pow(2).mean().sqrt()), float(X.max()), 100 * float((X == 0).float().mean()), float(W.pow(2).mean().sqrt()), float(W.abs().max())))
    Xf = X.permute(1, 0, 2, 3).reshape(K, N)                                    # [K][N], n = img * npix + pix
    ref = W.double() @ Xf.double()
    rms_ref = float(ref.pow(2).mean().sqrt())
    flops = 2.0 * M * N * K

    def err(C):
        d = C.double() - ref
        return float(d.abs().max()) / rms_ref, float(d.pow(2).mean().sqrt()) / rms_ref

    # ---- shipped fp32 MFMA kernel ----
    wT, ktab, lay = ops.pack_weights(W.view(M, K, 1, 1))
    out = torch.empty(imgs, M, h, w, device=DEV)
    us32 = timed(lambda: ops.conv2d(X, wT, M, 1, 1, 0, ktab=ktab, out=out, w_layout=lay, tile=4))
    C32 = out.permute(1, 0, 2, 3).reshape(M, N)
    e32 = err(C32)
    print('%-34s %8.1f us  %7.1f TFLOP/s   max err %.3e  rms err %.3e   (relative to rms |C| = %.3f)' %
          ('fp32 MFMA (shipped, 16x16x4 f32)', us32, flops / us32 / 1e6, e32[0], e32[1], rms_ref))
    # fp32 result computed by torch (rocBLAS sgemm) for orientation
    Ct = W @ Xf
    et = err(Ct)
    print('%-34s %8s     %7s            max err %.3e  rms err %.3e' % ('fp32 torch matmul (rocBLAS)', '', '', et[0], et[1]))

    # ---- bf16 x 3 ----
    Wp = torch.empty(3 * (K // 8) * M * 8, dtype=torch.int16, device=DEV)
    Xp = torch.empty(3 * (K // 8) * N * 8, dtype=torch.int16, device=DEV)
    assert L.bf16x3_split_w(P(W.data_ptr()), M, K, P(Wp.data_ptr()), st) == 0
    us_split = timed(lambda: L.bf16x3_split_act(P(X.data_ptr()), imgs, K, npix, P(Xp.data_ptr()), st))
    print('%-34s %8.1f us   (fp32 NCHW -> 3 bf16 planes [K/8][N][8]; in a pipeline this is the producing conv\'s epilogue: 6 instead of 4 bytes per activation)'
          % ('split of the activations', us_split))
    C = torch.empty(M, N, device=DEV)
    res = {}
    for np_, name in ((6, 'bf16 x3, 6 products'), (106, 'bf16 x3, 6 products, second form'), (9, 'bf16 x3, all 9 products'), (109, 'bf16 x3, 9 products, second form'), (3, 'bf16 x2-like, 3 products (hi, mid)'), (1, 'plain bf16, 1 product')):
        C.zero_()
        rc = L.bf16x3_gemm(np_, P(Wp.data_ptr()), P(Xp.data_ptr()), P(C.data_ptr()), M, N, K, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        e = err(C)
        us = timed(lambda: L.bf16x3_gemm(np_, P(Wp.data_ptr()), P(Xp.data_ptr()), P(C.data_ptr()), M, N, K, st))
        note = '' if np_ in (6, 9, 106, 109) else '   (loads all three planes: rate not representative)'
        res[np_] = (us, e)
        print('%-34s %8.1f us  %7.1f TFLOP/s   max err %.3e  rms err %.3e%s' % (name, us, flops / us / 1e6, e[0], e[1], note))
    us6, e6 = min((res[6], res[106]), key=lambda v: v[0])
    print('# the 6-product form against the shipped fp32-MFMA kernel: %.2fx the rate (%.2fx with the split pass counted), %.2fx the rms error, %.2fx the max error'
          % (us32 / us6, us32 / (us6 + us_split), e6[1] / e32[1], e6[0] / e32[0]))


if __name__ == '__main__':
    main()
