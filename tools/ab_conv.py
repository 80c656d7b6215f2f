"""A/B timing of several libsnnhip.so builds in ONE process, interleaved rounds (cdna_hip_programming.md §5.4 rule 24):
fused conv+LIF (Cityscapes pyramid, b=2, T=8) and fc6+LIF (2000 RoIs, T=12) through the raw C ABI.
usage: python tools/ab_conv.py name1=/path/lib1.so name2=/path/lib2.so ... [--rounds 7] [--check]"""
import ctypes as C
import statistics
import sys

import torch

sys.path.insert(0, ".")
from snn_automotive_object_detection_amd import _lib, ops

libs = []
rounds, check = 7, False
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--rounds":
        rounds = int(args.pop(0))
    elif a == "--check":
        check = True
    else:
        name, path = a.split("=", 1)
        env = {}
        if ":" in path:                                  # name=path:VAR=VAL,VAR=VAL  (debug knobs of that build, frozen at load)
            path, kv = path.split(":", 1)
            env = dict(x.split("=") for x in kv.split(","))
        import os, shutil, tempfile
        if any(path == q for _, q in [(n, getattr(l, "_path", None)) for n, l in libs]):     # same file twice: dlopen would share it
            cp = tempfile.mktemp(suffix=".so"); shutil.copy(path, cp); path_load = cp
        else:
            path_load = path
        os.environ.update(env)
        lib = C.CDLL(path_load)
        lib._path = path
        lib._unpadded = name.startswith("R1")            # round-1 builds read un-padded encoder planes
        lib._clock = "CLK" in name                       # -DSNN_EXP_CLOCK build: stamps behind the spike planes
        lib._wm = env.get("SNN_STAGE_PLANES") == "wm"    # this instance takes its input planes word-major [T][word][row]
        if hasattr(lib, "snn_debug_reload_knobs"):
            lib.snn_debug_reload_knobs()
        for k in env:
            os.environ.pop(k)
        for n, (res, at) in (list(_lib.SYMBOLS.items()) + list(_lib.DEBUG_SYMBOLS.items())):
            if hasattr(lib, n):
                getattr(lib, n).restype = res
                getattr(lib, n).argtypes = at
        libs.append((name, lib))
dev = torch.device("cuda:0")
torch.manual_seed(0)
p = ops.make_params(ops.LIFParameters(v_th=torch.tensor(0.25)), ops.LIFParameters(alpha=100, v_th=torch.tensor(0.1)))
LEVELS = [(192, 384), (96, 192), (48, 96), (24, 48), (12, 24)]
T = 8
feats = [torch.randn(2, 256, h, w, device=dev) * 1.7 for h, w in LEVELS]          # encoder rate ~0.2-0.3 like the backbone-fed bench
encs = torch.cat([ops.encode_nchw(f, T, p) for f in feats], dim=1).contiguous()
encs_pad = ops.pad_planes(encs, [(2, h, w) for h, w in LEVELS])        # zero-halo planes (builds from round 2 on read these)
PP = encs_pad.shape[1]
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
wb = ops.pack_conv3x3_bf16x3(w)
shapes = [(2, h, w_) for h, w_ in LEVELS]
P = encs.shape[1]
lv = (_lib.snn_rpn_level * 5)(*[_lib.snn_rpn_level(None, n, h, w_, 0) for n, h, w_ in shapes])
N_WG_MAX = 40000
spk_buf = torch.zeros((T * P * 8 + N_WG_MAX * 4,), dtype=torch.int32, device=dev)      # + room for 2 x uint64 clock stamps per work-group
spk = spk_buf[:T * P * 8].view(T, P, 8)
R, D, Hd, T6 = 2000, 12544, 1024, 12
x = torch.randn(R, D, device=dev) * 1.7
enc6 = ops.encode_rows(x, T6, p)
w6 = torch.randn(Hd, D, device=dev) / D ** 0.5
w6b = ops.pack_linear_bf16x3(w6)
spk6_buf = torch.zeros((T6 * R * (Hd // 32) + N_WG_MAX * 4,), dtype=torch.int32, device=dev)
spk6 = spk6_buf[:T6 * R * (Hd // 32)].view(T6, R, Hd // 32)
st = torch.cuda.current_stream().cuda_stream


encs_pad_wm = encs_pad.permute(0, 2, 1).contiguous()
enc6_wm = None


def conv(lib):
    e, n = (encs, P) if getattr(lib, "_unpadded", False) else ((encs_pad_wm, PP) if getattr(lib, "_wm", False) else (encs_pad, PP))
    rc = lib.snn_conv3x3_lif_bf16x3(e.data_ptr(), n * 8, lv, 5, 256, 256, T, C.byref(p), wb.data_ptr(), spk.data_ptr(), P * 8, st)
    assert rc == 0, lib.snn_last_error()


def fc6(lib):
    global enc6_wm
    if getattr(lib, "_wm", False) and enc6_wm is None:
        enc6_wm = enc6.permute(0, 2, 1).contiguous()
    a = enc6_wm if getattr(lib, "_wm", False) else enc6
    rc = lib.snn_spike_gemm_lif_bf16x3(a.data_ptr(), T6, R, D, Hd, C.byref(p), w6b.data_ptr(), spk6.data_ptr(), R * (Hd // 32), st)
    assert rc == 0, lib.snn_last_error()


def tm(fn, lib, n=6):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(lib); b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


ref = {}
for name, lib in libs:                       # warm-up + (optional) equality of the spike planes with the first build's
    conv(lib); fc6(lib); torch.cuda.synchronize()
    if check:
        if not ref:
            ref = {"c": spk.clone(), "f": spk6.clone()}
        else:
            print("check %-12s conv planes equal: %s   fc6 planes equal: %s" % (name, torch.equal(spk, ref["c"]), torch.equal(spk6, ref["f"])))
res = {name: {"conv": [], "fc6": []} for name, _ in libs}
for r in range(rounds):
    for name, lib in libs:
        res[name]["conv"].append(tm(conv, lib))
        res[name]["fc6"].append(tm(fc6, lib))
def clock_of(buf, n_words):
    st_ = buf[n_words:].view(torch.int64).view(-1, 2).cpu()
    st_ = st_[(st_[:, 1] > 0)]
    if len(st_) == 0:
        return None
    ghz = (st_[:, 0].double() / st_[:, 1].double()) * 0.1
    return float(ghz.median()), float(st_[:, 0].double().median())


for name, lib in libs:
    if getattr(lib, "_clock", False):
        for _ in range(40):                                  # sustained load, then read the stamps of the last launch
            conv(lib)
        torch.cuda.synchronize()
        c = clock_of(spk_buf, T * P * 8)
        for _ in range(80):
            fc6(lib)
        torch.cuda.synchronize()
        f = clock_of(spk6_buf, T6 * R * (Hd // 32))
        print("%-14s in-kernel clock (median over work-groups): conv loop %.3f GHz, %.0f cycles per tile   fc6 loop %.3f GHz, %.0f cycles per tile" % (
            name, c[0], c[1], f[0], f[1]))
base = None
for name, _ in libs:
    c, f = res[name]["conv"], res[name]["fc6"]
    mc, mf = statistics.median(c), statistics.median(f)
    if base is None:
        base = (mc, mf)
    print("%-14s conv+LIF median %.4f ms (min %.4f)  %+5.1f %%    fc6+LIF median %.4f ms (min %.4f)  %+5.1f %%" % (
        name, mc, min(c), (mc / base[0] - 1) * 100, mf, min(f), (mf / base[1] - 1) * 100))
