import sys, time, torch
sys.path.insert(0, '.')
from snn_automotive_object_detection_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev)
# fc6
R, D, Hd, T = 2000, 12544, 1024, 12
T6 = T
x = torch.randn(R, D, device=dev)
enc = ops.encode_rows(x, T, p)
w6 = torch.randn(Hd, D, device=dev) / D ** 0.5
w6p = ops.pack_linear(w6); w6b = ops.pack_linear_bf16x3(w6)
t1 = tm(lambda: ops.spike_gemm(enc.view(T * R, -1), D, Hd, w6p))
t2 = tm(lambda: ops.spike_gemm_bf16x3(enc.view(T * R, -1), D, Hd, w6b))
fl = 2.0 * T * R * D * Hd
print('fc6  f32 %.3f ms (%.1f TF)   bf16x3 %.3f ms (%.1f TF algorithmic, %.1f TF executed)' % (t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, 3 * fl / t2 / 1e9))
# conv
LEVELS = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]
feats = [torch.randn(2, 256, h, w, device=dev) for h, w in LEVELS]
T = 8
encs = torch.cat([ops.encode_nchw(f, T, p) for f in feats], dim=1).contiguous()
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
wb = ops.pack_conv3x3_bf16x3(w)
shapes = [(2, h, w_) for h, w_ in LEVELS]
P = encs.shape[1]
t3 = tm(lambda: ops.spike_conv3x3_bf16x3(encs, shapes, 256, 256, wb), 3)
cur = ops.spike_conv3x3_bf16x3(encs, shapes, 256, 256, wb)
t4 = tm(lambda: ops.lif_scan(cur, 256, p), 3)
fl = 2.0 * T * P * 9 * 256 * 256
print('conv bf16x3 gemm %.3f ms (%.1f TF algorithmic, %.1f TF executed)  + lif_scan %.3f ms  [fused f32 kernel: 13.1 ms]' % (t3, fl / t3 / 1e9, 3 * fl / t3 / 1e9, t4))
t5 = tm(lambda: ops.conv3x3_lif_bf16x3(encs, shapes, 256, 256, p, wb), 3)
print('conv bf16x3 FUSED conv+LIF %.3f ms   (un-fused pair %.3f ms)' % (t5, t3 + t4))
# ---- fp4 x fp6 block-scaled path (k_gemm_mx)
w6m = ops.pack_linear_mx(w6)
t6 = tm(lambda: ops.spike_gemm_mx(enc.view(T6 * R, -1), D, Hd, w6m))
fl6 = 2.0 * T6 * R * D * Hd
print('fc6  mxfp6 %.3f ms (%.1f TF algorithmic)' % (t6, fl6 / t6 / 1e9))
wm = ops.pack_conv3x3_mx(w)
def pad_planes(pl, shapes):          # zero halo around every image (the mx conv's input format)
    out, pos = [], 0
    for n, h, w_ in shapes:
        blk = pl[:, pos:pos + n * h * w_].reshape(pl.shape[0], n, h, w_, -1)
        out.append(torch.nn.functional.pad(blk, (0, 0, 1, 1, 1, 1)).reshape(pl.shape[0], n * (h + 2) * (w_ + 2), -1))
        pos += n * h * w_
    return torch.cat(out, dim=1).contiguous()
encs_p = pad_planes(encs, shapes)
t7 = tm(lambda: ops.spike_conv3x3_mx(encs_p, shapes, 256, 256, wm), 3)
t8 = tm(lambda: ops.conv3x3_lif_mx(encs_p, shapes, 256, 256, p, wm), 3)
print('conv mxfp6 gemm %.3f ms (%.1f TF algorithmic)   FUSED conv+LIF %.3f ms' % (t7, fl / t7 / 1e9, t8))
