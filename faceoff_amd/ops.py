"""Host-side operator layer: torch tensors in (device memory + stream plumbing only), C-ABI calls out.

Activations are channels-last views [N,H,W,C] whose last-dim stride is 1 and whose pixel
stride `ld` may exceed C (channel slices of a wider buffer = free torch.cat).
Filters stay in the reference's checkpoint layouts and are packed on the GPU each step.
"""
from __future__ import annotations

import ctypes as C
import os as _os

import torch

from . import _lib
from ._lib import ConvDesc, FO_IN_RELU, FO_BIAS, FO_MASK, FO_ADD, FO_OUT_RELU, FO_DEPTH2SPACE, FO_OUT_F32  # noqa: F401


class KernelProfiler:
    """Live per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    Launches are grouped by the SYMBOL of the kernel the library launched -- name and template arguments as rocprofv3 prints them, reported
    by the library itself (fo_kernel_notes / fo_last_kernel) -- so that an entry of bench.py's JSON line can be looked up in
    profiles/*_kernel_stats.md.  `flops` is the ALGORITHMIC work of the launch: multiply-adds whose operands exist.  For a Conv3d that
    excludes the temporal taps that fall into clip padding (structural zeros the kernels skip: 2 of 15 at T=5, `temporal_share`);
    `nominal` is the padded-tap count SURVEY.md section 8(d) quotes.  An entry point that launches a main kernel plus a small reduce
    (filter gradients) is timed as a whole under the main kernel's symbol."""

    def __init__(self, detail=False):
        self.records = {}     # symbol -> list of (start_event, end_event, flops)
        self._cur = None
        self.detail = detail  # key launches by geometry as well (tools/, not bench.py's JSON line)
        self._lib = _lib.load()
        self._lib.fo_kernel_notes(1)

    def close(self):
        self._lib.fo_kernel_notes(0)

    def begin(self, label, flops, nominal=None):
        """label: the call's own description; used as the key only if the library noted no kernel.  In detail mode a trailing
        " [geometry]" of the label is appended to the symbol."""
        self._lib.fo_last_kernel()           # (reading clears: a stale note of an un-profiled call must not name this one)
        s = torch.cuda.Event(enable_timing=True)
        s.record(torch.cuda.current_stream())
        self._cur = (label, s, (flops, flops if nominal is None else nominal))

    def end(self):
        label, s, flops = self._cur
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        sym = self._lib.fo_last_kernel().decode()
        name = sym or label
        if sym and self.detail and " [" in label:
            name = sym + label[label.index(" ["):]
        self.records.setdefault(name, []).append((s, e, flops))

    def summary(self):
        """symbol -> dict(launches, total_ms, avg_ms, flops_per_launch, tflops) (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f[0] for _, _, f in recs)
            fn = sum(f[1] for _, _, f in recs)
            out[name] = dict(launches=len(recs), total_ms=ms, avg_ms=ms / len(recs), flops_per_launch=fl / len(recs),
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                             tflops_nominal=fn / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out


PROFILER = None   # set to a KernelProfiler to time conv launches
LEDGER = None     # set by StepLedger.open(): _ptr() then notes every tensor handed to the library


class StepLedger:
    """EVERY C-ABI call of a step with what bounds it (bench.py's step-level attainable floor): per call the entry point, the
    kernel symbol the library noted, the ALGORITHMIC bytes -- the distinct tensors (views: their own elements) handed to the call, each counted
    once, scratch workspaces excluded; an in-place operand counts once, so the figure errs low -- the algorithmic FLOP where the wrapper declares
    them (the KernelProfiler.begin() protocol: a ledger is installed as ops.PROFILER too), and HIP events around the call on its stream.

        floor(call) = max(bytes / HBM_BW, flop / (peak(dtype) * held_clock / 2.4 GHz))

    is what that launch could take at best given its own bound; bench.py sums it over the step.  Use with the side streams folded
    (engine.set_stream_overlap(False)) so that the events bracket a launch that has the GPU to itself."""

    def __init__(self):
        self.calls = []          # dict(entry, symbol, bytes, flops, ev0, ev1)
        self._touched = {}
        self._frac = {}
        self._pending = None
        self.detail = False
        self._lib = _lib.load()
        self._orig_call = None

    def open(self):
        global LEDGER, PROFILER
        self._lib.fo_kernel_notes(1)
        self._orig_call = _lib.call
        LEDGER = PROFILER = self
        _lib.call = self._call
        return self

    def close(self):
        global LEDGER, PROFILER
        _lib.call = self._orig_call
        LEDGER = PROFILER = None
        self._lib.fo_kernel_notes(0)

    # -- the KernelProfiler protocol the wrappers speak: flops of the call(s) that follow
    def begin(self, label, flops, nominal=None):
        self._pending = float(flops)

    def end(self):
        self._pending = None

    def touch(self, t):
        if t.is_cuda:
            self._touched[(t.data_ptr(), int(t.numel() * t.element_size() * self._frac.pop(t.data_ptr(), 1.0)))] = True

    def fraction(self, t, f):
        """The NEXT call touches only the fraction f of tensor t (a sub-pixel phase of a transposed convolution writes every other pixel of every
        other row of the output it is handed: a quarter)."""
        if t is not None:
            self._frac[t.data_ptr()] = f

    def _call(self, name, *args):
        scratch = {b.data_ptr() for b in _ws_cache.values()}
        nbytes = sum(n for (ptr, n) in self._touched if ptr not in scratch)
        self._touched = {}
        flops, self._pending = self._pending or 0.0, None
        self._lib.fo_last_kernel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        e0.record(st)
        self._orig_call(name, *args)
        e1.record(st)
        self._frac = {}
        self.calls.append(dict(entry=name, symbol=self._lib.fo_last_kernel().decode() or name, bytes=nbytes, flops=flops, ev0=e0, ev1=e1))

    def floor(self, steps, hbm_bw, peak_of, held_clock_of):
        """-> dict for bench.py's JSON line.  peak_of(symbol) -> dense MFMA TFLOP/s at 2.4 GHz; held_clock_of(symbol) -> GHz the chip holds under
        that kernel class (profiles/*_pmc.md); hbm_bw in B/s.  Synchronises."""
        torch.cuda.synchronize()
        tot = tb = tf = meas = over = 0.0
        rows = {}
        for c in self.calls:
            t_b = c["bytes"] / hbm_bw
            t_f = c["flops"] / (peak_of(c["symbol"]) * 1e12 * held_clock_of(c["symbol"]) / 2.4) if c["flops"] else 0.0
            fl = max(t_b, t_f)
            ms = c["ev0"].elapsed_time(c["ev1"])
            tot, tb, tf, meas = tot + fl, tb + t_b, tf + t_f, meas + ms
            over += max(0.0, fl * 1e3 - ms)              # a launch cannot beat its floor: anything here is the byte model counting too much
            r = rows.setdefault(c["symbol"], dict(launches=0, floor_ms=0.0, measured_ms=0.0, bound_hbm=0, bound_mfma=0))
            r["launches"] += 1
            r["floor_ms"] += fl * 1e3
            r["measured_ms"] += ms
            r["bound_hbm" if t_b >= t_f else "bound_mfma"] += 1
        k = 1.0 / steps
        gaps = sorted(rows.items(), key=lambda kv: -(kv[1]["measured_ms"] - kv[1]["floor_ms"]))
        return {"step_floor_ms": round(tot * 1e3 * k, 3), "step_floor_perfect_overlap_ms": round(max(tb, tf) * 1e3 * k, 3),
                "sum_hbm_ms": round(tb * 1e3 * k, 3), "sum_mfma_ms": round(tf * 1e3 * k, 3), "launches_per_step": round(len(self.calls) * k, 1),
                "measured_serial_launch_ms": round(meas * k, 3), "floor_above_measured_ms": round(over * k, 3),
                "largest_gaps": [{"kernel": n, "launches_per_step": round(r["launches"] * k, 1), "measured_ms": round(r["measured_ms"] * k, 3),
                                  "floor_ms": round(r["floor_ms"] * k, 3), "bound": "hbm" if r["bound_hbm"] >= r["bound_mfma"] else "mfma"}
                                 for n, r in gaps[:8]]}


_logged = set()


def _log_once(key, msg):
    """One line on stderr the first time a size-dependent downgrade is taken (never silent, never per step)."""
    if key not in _logged:
        _logged.add(key)
        import sys
        sys.stderr.write(msg + "\n")


def temporal_share(T, kd=3, pad=1):
    """Fraction of a Conv3d's (frame, depth tap) pairs that read a real frame of the clip: (3T-2)/3T for k=3, p=1."""
    valid = sum(1 for t in range(T) for k in range(kd) if 0 <= t + k - pad < T)
    return valid / float(T * kd)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if LEDGER is not None and t is not None:
        LEDGER.touch(t)
    return C.c_void_p(0 if t is None else t.data_ptr())


def ld_of(t, dtype=torch.float32):
    """Pixel stride (elements) of a channels-last view [N,H,W,C] (or [rows,C])."""
    assert t.dtype == dtype and t.is_cuda, f"{dtype} device tensor required"
    assert t.stride(-1) == 1, "channels must be contiguous"
    ld = t.stride(-2)
    if t.dim() == 4:
        n, h, w, c = t.shape
        assert t.stride(1) == w * ld and (n == 1 or t.stride(0) == h * w * ld), "rows must be dense"
    return ld


def dense_f32(t, what):
    """Loader-side tensors (NCHW images) go to the kernels as raw pointers: they must be dense fp32 device memory.
    A wrong dtype, a CPU tensor or a strided view would otherwise be read as garbage or fault on the GPU."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{what}: a float32 tensor on the GPU is required, got "
                        f"{getattr(t, 'dtype', type(t))} on {getattr(t, 'device', '?')}")
    return t if t.is_contiguous() else t.contiguous()


def pad_out(c):
    """Output-channel padding the conv kernel's N tile implies (see fo_conv_igemm)."""
    return (c + 127) // 128 * 128 if c > 64 else (64 if c > 32 else 32)


def pad_in(c):
    if c >= 32:
        assert c % 32 == 0
        return c
    return 8 if c <= 8 else 16


# ------------------------------------------------------------------ packing
def pack_conv(w, out=None):
    """[O][I][*taps] -> [Opad][taps][Ipad]"""
    O, I = w.shape[:2]
    taps = w[0, 0].numel()
    Op, Ip = pad_out(O), pad_in(I)
    if out is None:
        out = torch.empty(Op * taps * Ip, device=w.device, dtype=torch.float32)
    _lib.call("fo_pack_conv", _ptr(w), _ptr(out), O, I, taps, Op, Ip, _stream())
    return out


def pack_conv_dgrad(w, out=None):
    """stride-1 dgrad filter: [O][I][taps] -> [Ipad_as_out][taps reversed][Opad_as_in]"""
    O, I = w.shape[:2]
    taps = w[0, 0].numel()
    Op, Ip = pad_in(O), pad_out(I)
    if out is None:
        out = torch.empty(Ip * taps * Op, device=w.device, dtype=torch.float32)
    _lib.call("fo_pack_conv_dgrad", _ptr(w), _ptr(out), O, I, taps, Op, Ip, _stream())
    return out


def pack_convT(w, out=None):
    """[Ci][Co][4][4] -> [4][Copad][4][Cipad] (sub-pixel phases)"""
    Ci, Co = w.shape[:2]
    Cip, Cop = pad_in(Ci), pad_out(Co)
    if out is None:
        out = torch.empty(16 * Cop * Cip, device=w.device, dtype=torch.float32)
    _lib.call("fo_pack_convT_k4s2", _ptr(w), _ptr(out), Ci, Co, Cip, Cop, _stream())
    return out


CONVT_CELLS = not _os.environ.get("FACEOFF_NO_CONVT_CELLS")     # few-channel transposed conv: k2 cell form (K = 4 Ci) instead of 3x3 (9 Ci)


def pack_convT_fused(w, out=None):
    """[Ci][Co<=8][4][4] -> the transposed conv as one filter bank for FO_DEPTH2SPACE: [32 = 4 phases x 8][4 taps][Cipad] (cell
    form, default) or [32][9][Cipad] (3x3 form)"""
    Ci, Co = w.shape[:2]
    Cip = pad_in(Ci)
    taps = 4 if CONVT_CELLS else 9
    if out is None:
        out = torch.empty(32 * taps * Cip, device=w.device, dtype=torch.float32)
    _lib.call("fo_pack_convT_k4s2_cells" if CONVT_CELLS else "fo_pack_convT_k4s2_fused", _ptr(w), _ptr(out), Ci, Co, Cip, _stream())
    return out


# ------------------------------------------------------------------ conv launches
def _desc(**kw):
    d = ConvDesc()
    for k, v in kw.items():
        setattr(d, k, int(v))
    return d


def conv_igemm(x, wp, bias, out, *, T=1, k=(1, 3, 3), stride=1, pad=(0, 1, 1), cin=None, cout=None,
               flags=0, mask=None, add=None, ostride=1, oph=(0, 0), mgrid=None):
    """One fo_conv_igemm launch.  x/out/mask/add: channels-last views."""
    N, Hin, Win, _ = x.shape
    _, Hout, Wout, _ = out.shape
    Hm, Wm = mgrid if mgrid is not None else (Hout, Wout)
    if bias is not None:
        flags |= FO_BIAS
    if mask is not None:
        flags |= FO_MASK
    if add is not None:
        flags |= FO_ADD
    d = _desc(N=N, T=T, Hin=Hin, Win=Win, Hm=Hm, Wm=Wm, Hout=Hout, Wout=Wout,
              Cin=cin if cin is not None else x.shape[-1], Cout=cout if cout is not None else out.shape[-1],
              KD=k[0], KH=k[1], KW=k[2], stride=stride, padD=pad[0], padH=pad[1], padW=pad[2],
              ostride=ostride, ophH=oph[0], ophW=oph[1], ldIn=ld_of(x), ldOut=ld_of(out),
              ldMask=ld_of(mask) if mask is not None else 0, ldAdd=ld_of(add) if add is not None else 0, flags=flags)
    prof = PROFILER
    if prof is not None:
        var = _lib.load().fo_conv_igemm_variant(C.byref(d))
        kname = "conv_igemm3" if var == 3 else "conv_igemm_bn%d%s" % (var, "_smallc" if d.Cin < 32 else "")
        nominal = 2.0 * N * Hm * Wm * d.Cout * (k[0] * k[1] * k[2] * d.Cin)
        flops = nominal * (temporal_share(T, k[0], pad[0]) if k[0] > 1 else 1.0)
        if prof.detail:
            kname += f" [{N}x{Hm}x{Wm} {d.Cin}->{d.Cout} k{k[0]}{k[1]}{k[2]} s{stride} f{flags}]"
        prof.begin(kname, flops, nominal)
    if LEDGER is not None and ostride > 1:          # a sub-pixel phase: every ostride-th pixel of every ostride-th row of out / mask / add
        for t in (out, mask, add):
            LEDGER.fraction(t, 1.0 / (ostride * ostride))
    _lib.call("fo_conv_igemm", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask), _ptr(add), _ptr(out), _stream())
    if prof is not None:
        prof.end()


def resblock_fwd(x, wp1, b1, wp3, b3, hbuf, out, out_relu):
    """One fo_resblock_fwd launch: hbuf = relu(conv3x3(relu(x)) + b1), out = [relu](conv1x1(hbuf) + b3 + x)."""
    N, H, W, Cc = x.shape
    d = _desc(N=N, T=1, Hin=H, Win=W, Hm=H, Wm=W, Hout=H, Wout=W, Cin=Cc, Cout=32, KD=1, KH=3, KW=3, stride=1, padD=0, padH=1, padW=1,
              ostride=1, ophH=0, ophW=0, ldIn=ld_of(x), ldOut=ld_of(hbuf), ldMask=0, ldAdd=ld_of(x), flags=0)
    prof = PROFILER
    if prof is not None:
        prof.begin("resblock_fwd" + (f" [{N}x{H}x{W} {Cc}->32->{Cc}]" if prof.detail else ""), 2.0 * N * H * W * 32 * (9 * Cc + Cc))
    _lib.call("fo_resblock_fwd", C.byref(d), _ptr(x), _ptr(wp1), _ptr(b1), _ptr(wp3), _ptr(b3), _ptr(hbuf), _ptr(out), ld_of(out),
              int(bool(out_relu)), _stream())
    if prof is not None:
        prof.end()


def resblock_bwd_conv3(g, hbuf, wp3, g_h, dw3, db3):
    """One pass over g (the ResBlock's output gradient) and hbuf (its hidden activation): g_h = (g W3) * (hbuf > 0), dw3 = g^T hbuf,
    db3 = column sums of g (fo_resblock_bwd_conv3)."""
    N, H, W, Cc = g.shape
    assert Cc == 128 and hbuf.shape[-1] == 32 and g_h.shape[-1] == 32
    M = N * H * W
    ws = _workspace(_lib.load().fo_resblock_bwd_conv3_ws_bytes(C.c_int64(M)), g.device)
    prof = PROFILER
    if prof is not None:
        prof.begin("resblock_bwd_conv3" + (f" [{N}x{H}x{W} 128->32 dgrad+wgrad+bias]" if prof.detail else ""), 2.0 * M * 128 * 32 * 2)
    _lib.call("fo_resblock_bwd_conv3", C.c_int64(M), _ptr(g), ld_of(g), _ptr(hbuf), ld_of(hbuf), _ptr(wp3), _ptr(g_h), ld_of(g_h), _ptr(dw3), _ptr(db3),
              _ptr(ws), C.c_int64(ws.numel() * 4), _stream())
    if prof is not None:
        prof.end()


# ------------------------------------------------------------------ Winograd F(m x m, 3x3) Conv3d, m = 2 or 4
def wino_tile(H, W, N=None):
    """Output-tile size for a Conv3d on HxW frames: 4 (4x fewer MFMA FLOP, fp32 error ~3e-6 of scale) when the plane
    stack can run as ONE banked GEMM launch (a plane's N * H/4 * W/4 rows are whole 128-row GEMM tiles and the 36 planes
    fit the 2 GiB buffer window), else 2 (2.25x fewer, error as the direct convolution), else 0 (odd sizes: direct)."""
    if H % 4 == 0 and W % 4 == 0 and N is not None and (N * (H // 4) * (W // 4)) % 128 == 0:
        if 36 * N * (H // 4) * (W // 4) * 128 * 4 < (1 << 31):
            return 4
        _log_once(("wino_tile", H, W),
                  f"faceoff_amd: {N} frames of {H}x{W} latents put the 36 F(4x4,3x3) planes past the 2 GiB buffer-descriptor window; "
                  f"these layers run as F(2x2,3x3) (2.25x instead of 4x fewer multiplies). Use at most "
                  f"{((1 << 31) - 1) // (36 * (H // 4) * (W // 4) * 128 * 4)} frames per step at this size to keep F(4x4).")
    return 2 if H % 2 == 0 and W % 2 == 0 else 0


def wino_filter(w, dgrad=False, m=2, out=None):
    """[O][I][KD][3][3] -> (m+2)^2 packed (3,1,1) filter banks [P][Opad][KD][Ipad] (dgrad: flipped taps, swapped channels)."""
    if w.dim() == 4:                       # Conv2d filter [O][I][3][3]: one depth tap
        w = w.unsqueeze(2)
    O, I, KD = w.shape[:3]
    rows, cols = (I, O) if dgrad else (O, I)
    Op, Ip = pad_out(rows), pad_in(cols)
    U = out if out is not None else torch.empty((m + 2) ** 2 * Op * KD * Ip, device=w.device, dtype=torch.float32)
    _lib.call("fo_wino_filter", _ptr(w.contiguous()), _ptr(U), O, I, KD, Op, Ip, int(dgrad), m, _stream())
    return U


_wino_cache = {}


def _wino_buffers(nfloats, device):
    """(V, M) scratch of the transformed planes, one pair per stream (see _workspace)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _wino_cache.get(key)
    if buf is None or buf[0].numel() < nfloats[0] or buf[1].numel() < nfloats[1]:
        n0 = max(nfloats[0], 0 if buf is None else buf[0].numel())
        n1 = max(nfloats[1], 0 if buf is None else buf[1].numel())
        buf = (torch.empty(n0, device=device, dtype=torch.float32), torch.empty(n1, device=device, dtype=torch.float32))
        _wino_cache[key] = buf
    # views of exactly the size asked for (the buffers are sized by the largest layer seen: ops.StepLedger counts a tensor's own bytes)
    return buf[0][:nfloats[0]], buf[1][:nfloats[1]]


# The Winograd-domain GEMMs on the bf16 matrix pipe (csrc/wino_gemm_split.hip: exact three-way bf16 split of both fp32 operands,
# six partial products, fp32 accumulation -- fp32 results to the last bit or two at 6/16 of the matrix time) or on the fp32 MFMA.
BF16X6 = bool(_os.environ.get("FACEOFF_BF16X6"))


_split_ws = {}


def wino_gemm(V, U, M, planes, N, T, P, cin, cout, kd):
    """The plane-stack GEMM M[xi] = V[xi] (x) U[xi] (csrc/wino_gemm.hip; csrc/wino_gemm_split.hip when BF16X6)."""
    if not BF16X6:
        _lib.call("fo_wino_gemm", _ptr(V), _ptr(U), _ptr(M), planes, N, T, P, cin, cout, kd, _stream())
        return
    need = 3 * planes * cout * kd * cin                       # bf16 elements: the filter banks as three planes
    key = (V.device, torch.cuda.current_stream(V.device).cuda_stream)
    ws = _split_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=V.device, dtype=torch.bfloat16)
        _split_ws[key] = ws
    _lib.call("fo_wino_gemm_split", _ptr(V), _ptr(U), _ptr(ws), _ptr(M), planes, N, T, P, cin, cout, kd, _stream())


def wino_wgrad_gemm_split(dM, V, planes, N, T, P, cin, cout, kd):
    """dU [planes][cout][cin][kd] by csrc/wino_wgrad_split.hip, or None where its shapes do not apply (BF16X6 only)."""
    if not BF16X6 or _os.environ.get("FACEOFF_NO_WGRAD_SPLIT") or cin % 128 or cout % 128 or (N * P) % 32 or (kd == 3 and P % 32):
        return None
    nbytes = _lib.load().fo_wino_wgrad_split_ws_bytes(planes, N, P, cin, cout, kd)
    if nbytes < 0:
        return None
    ws = _workspace(nbytes + planes * cout * cin * kd * 4 + 64, dM.device)
    dU = ws[(nbytes + 3) // 4 // 4 * 4 + 4:][:planes * cout * cin * kd]
    _lib.call("fo_wino_wgrad_split", _ptr(dM), _ptr(V), _ptr(dU), _ptr(ws), C.c_int64(nbytes), planes, N, T, P, cin, cout, kd, _stream())
    return dU


AFTER_GEMM = None      # hook called right after a Winograd-domain GEMM launch of a forward / data-gradient pass (engine: deferred wgrads)


def conv3d_winograd(x, U, bias, out, *, T, cin, cout, flags=0, mask=None, add=None, keep_v=False, m=2, kd=3):
    """Conv3d k3 p1 s1 (kd=3; or Conv2d 3x3 p1 s1, kd=1, T=1), or its data gradient with the dgrad filter banks, on
    [N,H,W,C] frames, clips of T frames.
    keep_v: return the transformed input planes in their own tensor (the filter gradient of the same layer needs exactly
    them: conv3d_wgrad_winograd(V=...)) instead of using the per-stream scratch."""
    N, H, W, _ = x.shape
    Ht, Wt, P = H // m, W // m, (m + 2) ** 2
    assert H % m == 0 and W % m == 0 and cin % 32 == 0
    if bias is not None:
        flags |= FO_BIAS
    if mask is not None:
        flags |= FO_MASK
    if add is not None:
        flags |= FO_ADD
    plane_v, plane_m = N * Ht * Wt * cin, N * Ht * Wt * cout
    V, M = _wino_buffers((0 if keep_v else P * plane_v, P * plane_m), x.device)
    if keep_v:
        V = torch.empty(P * plane_v, device=x.device, dtype=torch.float32)
    _lib.call("fo_wino_input", _ptr(x), ld_of(x), _ptr(V), N, H, W, cin, m, _stream())
    bank = pad_out(cout) * kd * cin                                 # floats per filter bank
    banked = (N * Ht * Wt) % 128 == 0
    per = max(1, min(P, ((1 << 31) - 1) // max(plane_v * 4, plane_m * 4))) if banked else 1   # planes per launch (2 GiB window)
    # fo_wino_gemm addresses the whole V stack through one buffer descriptor and reaches kd/2 frames past either end of it (those
    # offsets must still be below 2^31), and the split variant also needs the three bf16 filter-bank pieces inside one window:
    # the same predicate as the C side (csrc/wino_gemm.hip), so a stack just under 2 GiB takes the banked loop instead of FO_E_SHAPE
    one_launch = (banked and per == P and P * plane_v * 4 + 2 * (kd // 2) * Ht * Wt * cin * 4 < (1 << 31) and P * plane_m * 4 < (1 << 31)
                  and (not BF16X6 or 3 * P * pad_out(cout) * kd * cin * 2 < (1 << 31)))
    prof = PROFILER
    # every C2 shape: the whole plane stack as ONE launch of the persistent plane-stack GEMM kernel
    if one_launch and cin >= 64 and cout % 128 == 0 and not _os.environ.get("FACEOFF_NO_WINO_GEMM"):
        if prof is not None:
            nominal = 2.0 * P * N * Ht * Wt * cout * kd * cin
            prof.begin("wino_gemm" + (f" [F{m} {P}x{N}x{Ht}x{Wt} {cin}->{cout} k{kd}11]" if prof.detail else ""),
                       nominal * (temporal_share(T) if kd > 1 else 1.0), nominal)
        wino_gemm(V, U, M, P, N, T if kd > 1 else 1, Ht * Wt, cin, cout, kd)
        if AFTER_GEMM is not None:
            AFTER_GEMM()
        if prof is not None:
            prof.end()
        per = 0
    for p0 in (range(0, P, per) if per else ()):
        np_ = min(per, P - p0)
        d = _desc(N=np_ * N, T=T if kd > 1 else 1, Hin=Ht, Win=Wt, Hm=Ht, Wm=Wt, Hout=Ht, Wout=Wt, Cin=cin, Cout=cout, KD=kd, KH=1, KW=1,
                  stride=1, padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=cin, ldOut=cout, ldMask=0, ldAdd=0, flags=0)
        vin = V[p0 * plane_v:(p0 + np_) * plane_v]
        mout = M[p0 * plane_m:(p0 + np_) * plane_m]
        wp = U[p0 * bank:(p0 + np_) * bank]
        if prof is not None:
            nominal = 2.0 * np_ * N * Ht * Wt * cout * kd * cin
            # the same kernel instantiation as every other 128-column launch: one label, so that rocprofv3's per-kernel
            # averages and these events describe the same set of launches
            prof.begin("conv_igemm_bn%d" % (128 if cout > 64 else (64 if cout > 32 else 32))
                       + (f" [winograd F{m} GEMM {np_}x{N}x{Ht}x{Wt} {cin}->{cout} k{kd}11]" if prof.detail else ""),
                       nominal * (temporal_share(T) if kd > 1 else 1.0), nominal)
        if banked:
            _lib.call("fo_conv_igemm_banked", C.byref(d), _ptr(vin), _ptr(wp), _ptr(mout), N, _stream())
        else:
            _lib.call("fo_conv_igemm", C.byref(d), _ptr(vin), _ptr(wp), None, None, None, _ptr(mout), _stream())
        if prof is not None:
            prof.end()
    if mask is not None and _DIAG_WINO_NO_MASK:       # DIAGNOSTIC (wrong gradients, timing only): the data gradient's output transform without its
        mask, flags = None, flags & ~FO_MASK            # ReLU-mask read -- the upper bound of what a bit-plane mask could save
    _lib.call("fo_wino_output", _ptr(M), _ptr(bias), _ptr(mask), ld_of(mask) if mask is not None else 0, _ptr(add),
              ld_of(add) if add is not None else 0, _ptr(out), ld_of(out), N, H, W, cout, flags, m, _stream())
    return V if keep_v else None


# ---------------------------------------------------------------- Winograd F(4x4, 2x2) for the k4 s2 p1 stems (csrc/wino42.hip)
_DIAG_WINO_NO_MASK = bool(_os.environ.get("FACEOFF_DIAG_WINO_NO_MASK"))
W42 = not _os.environ.get("FACEOFF_NO_W42")
W42_MIN_ROWS = 1024          # tiles per plane below which the direct kernels are used (tests lower it to reach this path at small sizes)


def _rows128(n):
    return (n + 127) // 128 * 128


def w42_filter(w, transposed, out=None):
    """w [O][I][4][4] (Conv2d; or ConvTranspose2d [I_T][O_T][4][4] read as O := I_T, I := O_T) -> U [25][O][4I] (conv form) or
    [25][4I][O] (transposed form)."""
    O, I = w.shape[:2]
    U = out if out is not None else torch.empty(25 * O * 4 * I, device=w.device, dtype=torch.float32)
    _lib.call("fo_w42_filter", _ptr(w), _ptr(U), O, I, int(bool(transposed)), _stream())
    return U


def w42_conv_ok(N, H, W, cin, cout):
    """conv form (Conv2d k4 s2 p1 forward; data gradient of ConvTranspose2d k4 s2 p1) on [N,H,W,cin] -> [N,H/2,W/2,cout]"""
    rows = N * (H // 8) * (W // 8)
    return (W42 and H % 8 == 0 and W % 8 == 0 and cin % 16 == 0 and 4 * cin >= 64 and cout % 128 == 0 and rows >= W42_MIN_ROWS
            and 25 * _rows128(rows) * max(4 * cin, cout) * 4 < (1 << 31))


def w42_convT_ok(N, h, w, cin, cout):
    """transposed form (ConvTranspose2d k4 s2 p1 forward; data gradient of Conv2d k4 s2 p1) on [N,h,w,cin] -> [N,2h,2w,cout]"""
    rows = N * ((h + 4) // 4) * ((w + 4) // 4)
    return (W42 and cin % 32 == 0 and cin >= 64 and (4 * cout) % 128 == 0 and rows >= W42_MIN_ROWS
            and 25 * _rows128(rows) * max(cin, 4 * cout) * 4 < (1 << 31))


def _w42_gemm(V, U, M, rows, K, Nc, label):
    prof = PROFILER
    if prof is not None:
        nominal = 2.0 * 25 * rows * K * Nc
        prof.begin("wino_gemm" + (f" [F(4,2) 25x{rows} {K}->{Nc} {label}]" if prof.detail else ""), nominal, nominal)
    wino_gemm(V, U, M, 25, 1, 1, rows, K, Nc, 1)
    if AFTER_GEMM is not None:
        AFTER_GEMM()
    if prof is not None:
        prof.end()


def _epi(bias, mask, add, flags):
    return flags | (FO_BIAS if bias is not None else 0) | (FO_MASK if mask is not None else 0) | (FO_ADD if add is not None else 0)


def conv_k4s2_winograd(x, U, bias, out, *, cin, cout, flags=0, mask=None, add=None, keep_v=False):
    """Conv2d k4 s2 p1 (cin -> cout) on x [N,H,W,>=cin] into out [N,H/2,W/2,>=cout] as F(4x4, 2x2) over 2x2 pixel cells.
    U = w42_filter(w, False).  keep_v: return the transformed input (the layer's filter gradient needs exactly it)."""
    N, H, W, _ = x.shape
    rows = _rows128(N * (H // 8) * (W // 8))
    K = 4 * cin
    V, M = _wino_buffers((0 if keep_v else 25 * rows * K, 25 * rows * cout), x.device)
    if keep_v:
        # the rows that pad a plane to whole 128-row GEMM tiles must be ZERO in a kept V: the filter gradient contracts over all rows
        # of the plane (0 x uninitialised memory can be NaN)
        V = (torch.empty if rows == N * (H // 8) * (W // 8) else torch.zeros)(25 * rows * K, device=x.device, dtype=torch.float32)
    _lib.call("fo_w42_input_cells", _ptr(x), ld_of(x), _ptr(V), N, H, W, cin, C.c_int64(rows), _stream())
    _w42_gemm(V, U, M, rows, K, cout, "conv")
    _lib.call("fo_w42_output", _ptr(M), C.c_int64(rows), _ptr(bias), _ptr(mask), ld_of(mask) if mask is not None else 0, _ptr(add),
              ld_of(add) if add is not None else 0, _ptr(out), ld_of(out), N, H // 2, W // 2, cout, _epi(bias, mask, add, flags), _stream())
    return V if keep_v else None


def convT_k4s2_winograd(g, U, bias, out, *, cin, cout, flags=0, mask=None, add=None):
    """ConvTranspose2d k4 s2 p1 (cin -> cout) on g [N,h,w,>=cin] into out [N,2h,2w,>=cout] -- equally the data gradient of a
    Conv2d k4 s2 p1 (cout -> cin).  U = w42_filter(w, True) with w read as [cin][cout][4][4]."""
    N, h, w, _ = g.shape
    rows = _rows128(N * ((h + 4) // 4) * ((w + 4) // 4))
    V, M = _wino_buffers((25 * rows * cin, 25 * rows * 4 * cout), g.device)
    _lib.call("fo_w42_input_full", _ptr(g), ld_of(g), _ptr(V), N, h, w, cin, C.c_int64(rows), _stream())
    _w42_gemm(V, U, M, rows, cin, 4 * cout, "convT")
    _lib.call("fo_w42_output_cells", _ptr(M), C.c_int64(rows), _ptr(bias), _ptr(mask), ld_of(mask) if mask is not None else 0, _ptr(add),
              ld_of(add) if add is not None else 0, _ptr(out), ld_of(out), N, h, w, cout, _epi(bias, mask, add, flags), _stream())


def w42_wgrad_ok(N, H, W, cin, cout):
    """filter gradient of the conv form: x [N,H,W,cin] (cells), g [N,H/2,W/2,cout]"""
    rows = N * (H // 8) * (W // 8)
    return (W42 and H % 8 == 0 and W % 8 == 0 and cin % 16 == 0 and cout % 32 == 0 and rows % 32 == 0 and rows >= W42_MIN_ROWS
            and 25 * _rows128(rows) * max(4 * cin, cout) * 4 < (1 << 31))


def conv_k4s2_wgrad_winograd(x, g, dw, *, cin, cout, V=None, dbias=None):
    """dw [cout][cin][4][4] = filter gradient of Conv2d k4 s2 p1 from its input x [N,H,W,>=cin] and output gradient
    g [N,H/2,W/2,>=cout] (for a ConvTranspose2d: x := its output gradient, g := its input, dw its [I_T][O_T][4][4] weight).
    V: the forward's transformed input (conv_k4s2_winograd(keep_v=True)), rows padded to 128.  dbias (optional): column sums of g, formed
    inside g's transform."""
    N, H, W, _ = x.shape
    tiles = N * (H // 8) * (W // 8)
    K = 4 * cin
    if V is not None:
        rows = V.numel() // (25 * K)
        _, dM = _wino_buffers((0, 25 * rows * cout), x.device)
    else:
        rows = tiles
        V, dM = _wino_buffers((25 * rows * K, 25 * rows * cout), x.device)
        _lib.call("fo_w42_input_cells", _ptr(x), ld_of(x), _ptr(V), N, H, W, cin, C.c_int64(rows), _stream())
    if rows != tiles:
        dM[:25 * rows * cout].zero_()                                  # the padding rows of a plane contribute nothing
    if dbias is not None and cout % 4 == 0 and 256 % (cout // 4) == 0:
        bws = _workspace(_lib.load().fo_w42_gradout_bias_ws_bytes(N, H // 2, W // 2, cout), x.device)
        _lib.call("fo_w42_gradout_bias", _ptr(g), ld_of(g), _ptr(dM), N, H // 2, W // 2, cout, C.c_int64(rows), _ptr(dbias), _ptr(bws),
                  C.c_int64(bws.numel() * 4), _stream())
    else:
        _lib.call("fo_w42_gradout", _ptr(g), ld_of(g), _ptr(dM), N, H // 2, W // 2, cout, C.c_int64(rows), _stream())
        if dbias is not None:
            bias_grad(g, dbias, cout)
    # planes as frames of a (1,1,1) wgrad: one "frame" per plane with `rows` positions
    d = _desc(N=25, T=1, Hin=1, Win=rows, Hm=1, Wm=rows, Hout=1, Wout=rows, Cin=K, Cout=cout, KD=1, KH=1, KW=1, stride=1, padD=0, padH=0,
              padW=0, ostride=1, ophH=0, ophW=0, ldIn=K, ldOut=cout, ldMask=0, ldAdd=0, flags=0)
    nbytes = _lib.load().fo_wgrad_banked_ws_bytes(C.byref(d), 25)
    if nbytes < 0:
        _lib.check(-1, "fo_wgrad_banked_ws_bytes")
    ws = _workspace(nbytes + 25 * cout * K * 4 + 64, x.device)
    dU = ws[(nbytes + 3) // 4 // 4 * 4 + 4:][:25 * cout * K]
    prof = PROFILER
    if prof is not None:
        nominal = 2.0 * 25 * rows * cout * K
        prof.begin("conv_wgrad_%dx%d" % (cout, K) + (f" [F(4,2) GEMM 25x{rows}]" if prof.detail else ""), nominal, nominal)
    dUs = wino_wgrad_gemm_split(dM, V, 25, 1, 1, rows, K, cout, 1)
    if dUs is not None:
        dU = dUs
    else:
        _lib.call("fo_conv_wgrad_banked", C.byref(d), _ptr(dM), _ptr(V), _ptr(dU), cout, K, _ptr(ws), C.c_int64(nbytes), 25, _stream())
    if prof is not None:
        prof.end()
    _lib.call("fo_w42_wgrad_out", _ptr(dU), _ptr(dw), cout, cin, _stream())


def wino_wgrad_ok(H, W, N, T, m=2, kd=3):
    """The Winograd filter-gradient form needs frames that are multiples of m, (H/m * W/m) % 32 == 0 (the wgrad kernel's
    row-run walk over a plane flattened to one row per frame) and plane stacks inside the 2 GiB buffer window."""
    return ((T > 1 or kd == 1) and m in (2, 4) and H % m == 0 and W % m == 0 and ((H // m) * (W // m)) % 32 == 0 and N % T == 0
            and (m + 2) ** 2 * N * (H // m) * (W // m) * 128 * 4 < (1 << 31))


def _wgrad_winograd_dU(g, x, *, T, a_real, b_real, V=None, m=2, kd=3, dbias=None):
    """dU[xi][co][ci][kd] = sum over the frames of g / x of dM[xi] (x) V[xi]  (one banked wgrad GEMM launch); returns dU
    (a slice of this stream's workspace).  dbias (optional): receives the column sums of g, formed inside g's transform."""
    N, H, W, _ = x.shape
    Ht, Wt, P = H // m, W // m, (m + 2) ** 2
    cin, cout = b_real, a_real
    plane_v, plane_m = N * Ht * Wt * cin, N * Ht * Wt * cout
    if V is not None:                        # the forward pass kept its transformed input
        assert V.numel() == P * plane_v
        _, dM = _wino_buffers((0, P * plane_m), x.device)
    else:
        V, dM = _wino_buffers((P * plane_v, P * plane_m), x.device)
        _lib.call("fo_wino_input", _ptr(x), ld_of(x), _ptr(V), N, H, W, cin, m, _stream())
    if dbias is not None and 256 % (cout // 4) == 0:
        bws = _workspace(_lib.load().fo_wino_gradout_bias_ws_bytes(N, H, W, cout, m), x.device)
        _lib.call("fo_wino_gradout_bias", _ptr(g), ld_of(g), _ptr(dM), N, H, W, cout, m, _ptr(dbias), _ptr(bws), C.c_int64(bws.numel() * 4), _stream())
    else:
        _lib.call("fo_wino_gradout", _ptr(g), ld_of(g), _ptr(dM), N, H, W, cout, m, _stream())
        if dbias is not None:
            bias_grad(g, dbias, cout)
    d = _desc(N=P * N, T=T if kd > 1 else 1, Hin=1, Win=Ht * Wt, Hm=1, Wm=Ht * Wt, Hout=1, Wout=Ht * Wt, Cin=cin, Cout=cout, KD=kd,
              KH=1, KW=1, stride=1, padD=kd // 2, padH=0, padW=0, ostride=1, ophH=0, ophW=0, ldIn=cin, ldOut=cout, ldMask=0, ldAdd=0,
              flags=0)
    nbytes = _lib.load().fo_wgrad_banked_ws_bytes(C.byref(d), P)
    if nbytes < 0:
        _lib.check(-1, "fo_wgrad_banked_ws_bytes")
    ws = _workspace(nbytes + P * cout * cin * kd * 4 + 64, x.device)
    dU = ws[(nbytes + 3) // 4 // 4 * 4 + 4:][:P * cout * cin * kd]
    prof = PROFILER
    if prof is not None:
        nominal = 2.0 * P * N * Ht * Wt * cout * cin * kd
        prof.begin("conv_wgrad_%dx%d" % (cout, cin) + (f" [winograd F{m} GEMM {P}x{N}x{Ht}x{Wt} k{kd}11]" if prof.detail else ""),
                   nominal * (temporal_share(T) if kd > 1 else 1.0), nominal)
    dUs = wino_wgrad_gemm_split(dM, V, P, N, T if kd > 1 else 1, Ht * Wt, cin, cout, kd)
    if dUs is not None:
        dU = dUs
    else:
        _lib.call("fo_conv_wgrad_banked", C.byref(d), _ptr(dM), _ptr(V), _ptr(dU), cout, cin, _ptr(ws), C.c_int64(nbytes), P, _stream())
    if prof is not None:
        prof.end()
    return dU


def conv3d_wgrad_winograd(g, x, dw, dbias, *, T, a_real, b_real, V=None, m=2, kd=3):
    """Filter gradient of a Conv3d k3 p1 in the Winograd domain: dU[xi] = sum dM[xi] (x) V[xi] ((m+2)^2 banked wgrad
    GEMMs with a (3,1,1) geometry), dW = G^T dU G; 2.25x (m=2) / 4x (m=4) fewer MFMA FLOP than the direct form.
    dbias = column sums of g.  V: the forward's transformed input, if it was kept."""
    N = x.shape[0]
    cin, cout = b_real, a_real
    P = (m + 2) ** 2
    dU = _wgrad_winograd_dU(g, x, T=T, a_real=a_real, b_real=b_real, V=V, m=m, kd=kd, dbias=dbias)
    _lib.call("fo_wino_wgrad_out", _ptr(dU), _ptr(dw), cout, cin, kd, m, _stream())


# ------------------------------------------------------------------ bf16 conv family (LPIPS branch)
def pack_conv_bf16(w, taps_pad=None):
    """fp32 [O][I][*taps] -> bf16 [Opad][tapsPad][Ipad] (Ipad: multiple of 64, or 8 for the RGB layer)"""
    O, I = w.shape[:2]
    taps = w[0, 0].numel()
    Ip = 8 if I <= 8 else (I + 63) // 64 * 64
    tp = taps if taps_pad is None else taps_pad
    out = torch.empty(pad_out(O) * tp * Ip, device=w.device, dtype=torch.bfloat16)
    _lib.call("fo_pack_conv_bf16", _ptr(w), _ptr(out), O, I, taps, pad_out(O), Ip, tp, _stream())
    return out


def pack_conv_dgrad_bf16(w):
    """stride-1 dgrad filter: fp32 [O][I][taps] -> bf16 [Ipad_as_out][taps reversed][Opad_as_in]"""
    O, I = w.shape[:2]
    taps = w[0, 0].numel()
    Op, Ip = (O + 63) // 64 * 64, pad_out(I)
    out = torch.empty(Ip * taps * Op, device=w.device, dtype=torch.bfloat16)
    _lib.call("fo_pack_conv_dgrad_bf16", _ptr(w), _ptr(out), O, I, taps, Op, Ip, _stream())
    return out


def conv_bf16_pool_ok(N, H, W, cin, cout):
    """Can fo_conv_igemm_bf16_pool take this 3x3 same-size layer?  (the halo-tile kernel's geometry and size gate, csrc/conv_bf16.hip)"""
    import os
    if os.environ.get("FACEOFF_BF16_NO_HALO", "0") not in ("", "0") or os.environ.get("FACEOFF_NO_POOL_FUSION"):
        return False
    tiles = N * (H // 4) * (W // 32) * (cout // 64)
    forced = os.environ.get("FACEOFF_BF16_FORCE_HALO", "0") not in ("", "0")
    return cin == 64 and cout % 64 == 0 and H % 4 == 0 and W % 32 == 0 and (tiles >= 8 * _lib.cu_count() or forced)


def conv_bf16(x, wp, bias, out, *, k=(3, 3), stride=1, pad=(1, 1), cin=None, cout=None, flags=0, mask=None, pooled=None, pool_idx=None,
              mask_bits=None, out_bits=None, pooled_bits=None):
    """One fo_conv_igemm_bf16 launch.  x/out/mask: bf16 channels-last views; bias fp32.  pooled (optional, only where conv_bf16_pool_ok):
    receives the 2x2 max-pool of the result from the same launch, pool_idx (uint8 [N, H/2, W/2, Cout/4]) its arg-max codes.
    Bit planes (uint8 [N, H, W, Cout/8], bit c % 8 of byte c / 8 = value > 0; fo_conv_bf16_ex): mask_bits = the ReLU-backward mask instead of the
    tensor `mask`, out_bits / pooled_bits receive the planes of the result / of the pooled result."""
    N, Hin, Win, _ = x.shape
    _, Hout, Wout, _ = out.shape
    bf = torch.bfloat16
    if bias is not None:
        flags |= FO_BIAS
    if mask is not None or mask_bits is not None:
        flags |= FO_MASK
    d = _desc(N=N, T=1, Hin=Hin, Win=Win, Hm=Hout, Wm=Wout, Hout=Hout, Wout=Wout,
              Cin=cin if cin is not None else x.shape[-1], Cout=cout if cout is not None else out.shape[-1],
              KD=1, KH=k[0], KW=k[1], stride=stride, padD=0, padH=pad[0], padW=pad[1], ostride=1, ophH=0, ophW=0,
              ldIn=ld_of(x, bf), ldOut=ld_of(out, bf), ldMask=ld_of(mask, bf) if mask is not None else 0, ldAdd=0, flags=flags)
    prof = PROFILER
    if prof is not None:
        kname = "conv_bf16_bn%d%s" % (128 if d.Cout > 64 else (64 if d.Cout > 32 else 32), "_rgb" if d.Cin < 64 else "")
        flops = 2.0 * N * Hout * Wout * d.Cout * (k[0] * k[1] * d.Cin)
        if prof.detail:
            kname += f" [{N}x{Hout}x{Wout} {d.Cin}->{d.Cout} f{flags}]"
        prof.begin(kname, flops)
    if mask_bits is not None or out_bits is not None or pooled_bits is not None:
        assert pooled is None or mask is None
        ex = _lib.ConvExtra(pooled=_ptr(pooled).value, ldPooled=ld_of(pooled, bf) if pooled is not None else 0, pool_idx=_ptr(pool_idx).value,
                            mask_bits=_ptr(mask_bits).value, out_bits=_ptr(out_bits).value, pooled_bits=_ptr(pooled_bits).value)
        _lib.call("fo_conv_bf16_ex", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask), None, _ptr(out), C.byref(ex), _stream())
    elif pooled is not None:
        assert mask is None
        if pool_idx is not None:
            _lib.call("fo_conv_igemm_bf16_pool_idx", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(out), _ptr(pooled), ld_of(pooled, bf), _ptr(pool_idx),
                      _stream())
        else:
            _lib.call("fo_conv_igemm_bf16_pool", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(out), _ptr(pooled), ld_of(pooled, bf), _stream())
    else:
        _lib.call("fo_conv_igemm_bf16", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask), _ptr(out), _stream())
    if prof is not None:
        prof.end()


# ------------------------------------------------------------------ bf16-operand VQ-VAE step (csrc/conv_bf16.hip, wgrad_bf16.hip, bf16_ops.hip)
BF = torch.bfloat16


def to_bf16(x, out=None):
    """fp32 -> bf16 (round to nearest even) of a dense tensor or a channels-last view (last dim % 8 == 0)."""
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF)
    if x.dim() == 1:
        _lib.call("fo_f32_to_bf16", _ptr(x), C.c_int64(Cc), _ptr(out), C.c_int64(Cc), C.c_int64(1), Cc, _stream())
        return out
    rows = x.numel() // Cc
    _lib.call("fo_f32_to_bf16", _ptr(x), C.c_int64(ld_of(x)), _ptr(out), C.c_int64(ld_of(out, BF)), C.c_int64(rows), Cc, _stream())
    return out


def to_f32(x, out=None):
    Cc = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    rows = x.numel() // Cc
    _lib.call("fo_bf16_to_f32", _ptr(x), C.c_int64(ld_of(x, BF)), _ptr(out), C.c_int64(ld_of(out)), C.c_int64(rows), Cc, _stream())
    return out


def cat_nchw_to_nhwc8_bf16(a, b=None):
    """process_data's channel concatenation + layout change + rounding of the network input: [N,Ca,H,W] (+ [N,Cb,H,W]) fp32 -> bf16 [N,H,W,8]"""
    N, Ca, H, W = a.shape
    Cb = 0 if b is None else b.shape[1]
    assert Ca + Cb <= 8 and (b is None or b.shape == (N, Cb, H, W))
    y = torch.empty((N, H, W, 8), device=a.device, dtype=BF)
    _lib.call("fo_nchw2_to_nhwc8_bf16", _ptr(dense_f32(a, "source frames")), Ca, _ptr(None if b is None else dense_f32(b, "background frames")), Cb, _ptr(y),
              N, H, W, _stream())
    return y


def conv_bf16g(x, wp, bias, out, *, T=1, k=(1, 3, 3), stride=1, pad=(0, 1, 1), cin=None, cout=None, flags=0, mask=None, add=None, ostride=1,
               oph=(0, 0), mgrid=None):
    """One fo_conv_bf16 launch: bf16 x / wp / mask / add (channels-last views), fp32 bias; out bf16, or fp32 (FO_OUT_F32 is set from its dtype)."""
    N, Hin, Win, _ = x.shape
    _, Hout, Wout, _ = out.shape
    Hm, Wm = mgrid if mgrid is not None else (Hout, Wout)
    if bias is not None:
        flags |= FO_BIAS
    if mask is not None:
        flags |= FO_MASK
    if add is not None:
        flags |= FO_ADD
    if out.dtype == torch.float32:
        flags |= FO_OUT_F32
    d = _desc(N=N, T=T, Hin=Hin, Win=Win, Hm=Hm, Wm=Wm, Hout=Hout, Wout=Wout,
              Cin=cin if cin is not None else x.shape[-1], Cout=cout if cout is not None else out.shape[-1],
              KD=k[0], KH=k[1], KW=k[2], stride=stride, padD=pad[0], padH=pad[1], padW=pad[2], ostride=ostride, ophH=oph[0], ophW=oph[1],
              ldIn=ld_of(x, BF), ldOut=ld_of(out, out.dtype), ldMask=ld_of(mask, BF) if mask is not None else 0,
              ldAdd=ld_of(add, BF) if add is not None else 0, flags=flags)
    prof = PROFILER
    if prof is not None:
        nominal = 2.0 * N * Hm * Wm * d.Cout * (k[0] * k[1] * k[2] * d.Cin)
        same = stride == 1 and ostride == 1 and (Hm, Wm) == (Hout, Wout) == (Hin, Win) and not (flags & FO_DEPTH2SPACE)
        kname = "conv_bf16_big" if (same and d.Cout % 128 == 0 and d.Cin >= 32 and not (flags & FO_IN_RELU)) else (
            "conv_bf16_bn%d%s" % (128 if d.Cout > 64 else (64 if d.Cout > 32 else 32), "_c8" if d.Cin < 32 else ""))
        if prof.detail:
            kname += f" [{N}x{Hm}x{Wm} {d.Cin}->{d.Cout} k{k[0]}{k[1]}{k[2]} s{stride} f{flags}]"
        prof.begin(kname, nominal * (temporal_share(T, k[0], pad[0]) if k[0] > 1 else 1.0), nominal)
    if LEDGER is not None and ostride > 1:          # a sub-pixel phase: every ostride-th pixel of every ostride-th row of out / mask / add
        for t in (out, mask, add):
            LEDGER.fraction(t, 1.0 / (ostride * ostride))
    _lib.call("fo_conv_bf16", C.byref(d), _ptr(x), _ptr(wp), _ptr(bias), _ptr(mask), _ptr(add), _ptr(out), _stream())
    if prof is not None:
        prof.end()


def convT_fused_bf16(x, wpf, bias, out, *, cin, cout, flags=0):
    """k4 s2 p1 transposed conv with cout <= 8 as ONE launch (cell form, see convT_fused): out [N, 2H, 2W, >= 8], fp32 or bf16."""
    N, Hi, Wi, _ = x.shape
    conv_bf16g(x, wpf, bias, out, k=(1, 2, 2), stride=1, pad=(0, 1, 1), cin=cin, cout=32, flags=flags | FO_DEPTH2SPACE, mgrid=(Hi + 1, Wi + 1),
               oph=(1, cout))


def pack_convT_cells(w, out=None):
    """[Ci][Co][4][4] -> the cell form of the k4 s2 p1 transposed conv at any width: [4 phases x Cpp][4 taps][Cipad], Cpp = Co rounded up to 32
    (fo_pack_convT_k4s2_cells_n; fp32, rounded to bf16 by the caller)."""
    Ci, Co = w.shape[:2]
    Cip, Cpp = pad_in(Ci), (Co + 31) // 32 * 32
    if out is None:
        out = torch.empty(4 * Cpp * 4 * Cip, device=w.device, dtype=torch.float32)
    _lib.call("fo_pack_convT_k4s2_cells_n", _ptr(w), _ptr(out), Ci, Co, Cpp, Cip, _stream())
    return out


def convT_cells_bf16(x, wc, bias, out, *, cin, cout, flags=0, mask=None, add=None):
    """k4 s2 p1 transposed conv (or the dgrad of a k4 s2 p1 conv) as ONE launch: a k2 p1 conv over the (H+1) x (W+1) grid of 2x2-pixel cells whose
    4 Cpp GEMM columns are the four sub-pixel phases (FO_DEPTH2SPACE); the input is read once instead of once per phase.  mask / add at the output pixel."""
    N, Hi, Wi, _ = x.shape
    cpp = (cout + 31) // 32 * 32
    conv_bf16g(x, wc, bias, out, k=(1, 2, 2), stride=1, pad=(0, 1, 1), cin=cin, cout=4 * cpp, flags=flags | FO_DEPTH2SPACE, mask=mask, add=add,
               mgrid=(Hi + 1, Wi + 1), oph=(1, cout))


def convT_phases_bf16(x, wp4, bias, out, *, cin, cout, flags=0, mask=None, add=None):
    """k4 s2 p1 transposed conv (or the dgrad of a k4 s2 p1 conv) as 4 sub-pixel launches (bf16 operands)."""
    N, Hi, Wi, _ = x.shape
    per_phase = pad_out(cout) * 4 * cin
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        conv_bf16g(x, wp4[ph * per_phase:(ph + 1) * per_phase], bias, out, k=(1, 2, 2), stride=1, pad=(0, 1 - py, 1 - px), cin=cin, cout=cout,
                   flags=flags, mask=mask, add=add, ostride=2, oph=(py, px), mgrid=(Hi, Wi))


def conv_wgrad_bf16(P, Q, dw, dbias, *, T=1, k=(1, 3, 3), stride=1, pad=(0, 1, 1), a_real, b_real, in_relu=False):
    """dW[a][b][tap] (fp32) = sum_m P[m][a] Q[qpix(m,tap)][b] from bf16 P, Q; dbias (optional, fp32) = colsum(P)."""
    N, Hm, Wm, Ca = P.shape
    _, Hq, Wq, Cb = Q.shape
    d = _desc(N=N, T=T, Hin=Hq, Win=Wq, Hm=Hm, Wm=Wm, Hout=Hm, Wout=Wm, Cin=Cb, Cout=Ca, KD=k[0], KH=k[1], KW=k[2], stride=stride,
              padD=pad[0], padH=pad[1], padW=pad[2], ostride=1, ophH=0, ophW=0, ldIn=ld_of(Q, BF), ldOut=ld_of(P, BF), ldMask=0, ldAdd=0,
              flags=FO_IN_RELU if in_relu else 0)
    nbytes = _lib.load().fo_wgrad_bf16_ws_bytes(C.byref(d))
    if nbytes < 0:
        _lib.check(-1, "fo_wgrad_bf16_ws_bytes")
    ws = _workspace(nbytes, P.device)
    prof = PROFILER
    if prof is not None:
        wname = "wgrad_bf16_%dx%d" % (Ca, Cb)
        if prof.detail:
            wname += f" [{N}x{Hm}x{Wm} k{k[0]}{k[1]}{k[2]} s{stride}]"
        nominal = 2.0 * N * Hm * Wm * Ca * Cb * k[0] * k[1] * k[2]
        prof.begin(wname, nominal * (temporal_share(T, k[0], pad[0]) if k[0] > 1 else 1.0), nominal)
    _lib.call("fo_conv_wgrad_bf16", C.byref(d), _ptr(P), _ptr(Q), _ptr(dw), a_real, b_real, _ptr(dbias), _ptr(ws), C.c_int64(ws.numel() * 4), _stream())
    if prof is not None:
        prof.end()


def bias_grad_bf16(g, dbias, c_real):
    rows = g.shape[0] * g.shape[1] * g.shape[2]
    Cc = g.shape[-1]
    ws = _workspace(_lib.load().fo_bias_grad_bf16_ws_bytes(Cc), g.device)
    _lib.call("fo_bias_grad_bf16", _ptr(g), _ptr(dbias), C.c_int64(rows), Cc, c_real, ld_of(g, BF), _ptr(ws), _stream())


def _vq_stats(x, nvec, ind, stats, side):
    """EMA statistics (counts, esum) of an assignment; on `side` (a stream, behind the current stream's work) when given: nothing in the
    step waits for them -- they feed the codebook's EMA update, which the NEXT step reads."""
    def run():
        ws = _workspace(_lib.load().fo_vq_stats_ws_bytes(C.c_int64(nvec)), x.device)
        _lib.call("fo_vq_stats", _ptr(x), ld_of(x), C.c_int64(nvec), _ptr(ind), _ptr(stats[1:513]), _ptr(stats[513:]), _ptr(ws), _stream())
    if side is None:
        run()
        return
    side.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(side):
        run()
    # (no record_stream: the caller keeps x, ind and stats alive until the stream that made them has joined `side` -- VQVAEEngine does, in S)


import weakref as _weakref

_forced_seen = {}


def _forced(force_ind, shape, device):
    if force_ind is None:
        return None
    f = force_ind.to(device=device, dtype=torch.int64).contiguous()
    if tuple(f.shape) != tuple(shape):
        raise ValueError(f"faceoff_amd: forced code indices must be {tuple(shape)}, got {tuple(f.shape)}")
    # Range check: the kernel masks the index to [0, 512), so an out-of-range code (a -1 sentinel, say) would be a silently aliased code, never a wild
    # read -- and never an error either.  The check reads the tensor back (a host sync), so it runs ONCE per distinct tensor: the first
    # time a (storage, version) pair is seen -- teacher-forced parity runs hand the same tensor in step after step; FACEOFF_DEBUG=1 checks every call.
    key = (force_ind.data_ptr(), force_ind._version, tuple(force_ind.shape), str(force_ind.device))
    ref = _forced_seen.get(key)
    if _os.environ.get("FACEOFF_DEBUG") or ref is None or ref() is not force_ind:      # (the weak reference: a freed tensor's address may be handed to the next one)
        lo, hi = torch.aminmax(f)
        if int(lo) < 0 or int(hi) >= 512:
            raise ValueError("faceoff_amd: forced code indices must lie in [0, 512)")
        if len(_forced_seen) >= 16:
            _forced_seen.clear()
        _forced_seen[key] = _weakref.ref(force_ind)
    return f


def vq_assign_bf16out(x, embedT, enorm, q_f32, q_bf16, stats, train, stats_stream=None, force_ind=None):
    """vq_assign on the fp32 input x, writing the straight-through output both as fp32 (q_f32, kept for the backward) and as bf16
    (q_bf16: the operand of the next convolution; may be a channel slice)."""
    nvec = x.shape[0] * x.shape[1] * x.shape[2]
    ind = torch.empty(x.shape[:-1], device=x.device, dtype=torch.int64)
    f = _forced(force_ind, ind.shape, x.device)
    if LEDGER is not None and f is None:
        LEDGER.begin("vq_assign", 2.0 * nvec * 64 * 512)          # (the ledger's floor only: bench.py adds the VQ distance FLOP to its totals itself)
    _lib.call("fo_vq_assign2", _ptr(x), ld_of(x), C.c_int64(nvec), _ptr(embedT), _ptr(enorm), _ptr(ind), _ptr(q_f32), ld_of(q_f32), _ptr(stats[0:1]),
              _ptr(q_bf16), ld_of(q_bf16, BF), _ptr(f) if f is not None else None, _ptr(_workspace(4096, x.device)), _stream())
    if train:
        _vq_stats(x, nvec, ind, stats, stats_stream)
    return ind


def vq_bwd_bf16(gq, x, q, gdiff, gx):
    """gx (bf16) = bf16(gq (bf16) + gdiff * 2 / numel * (x - q)), x and q fp32"""
    nvec = x.shape[0] * x.shape[1] * x.shape[2]
    _lib.call("fo_vq_bwd_bf16", _ptr(gq), ld_of(gq, BF), _ptr(x), ld_of(x), _ptr(q), ld_of(q), _ptr(gdiff), C.c_float(2.0 / (nvec * 64)), _ptr(gx),
              ld_of(gx, BF), C.c_int64(nvec), _stream())


def convT_fused(x, wpf, bias, out, *, cin, cout, flags=0):
    """k4 s2 p1 transposed conv with cout <= 8 as ONE launch: 3x3 conv over the input grid, 32 GEMM columns =
    4 sub-pixel phases x 8 channels, depth-to-space in the epilogue (out: [N, 2H, 2W, >=8])."""
    N, Hi, Wi, _ = x.shape
    if CONVT_CELLS:      # k2 full correlation over the (Hi+1) x (Wi+1) cell grid; a cell's 4 pixels sit one row / column up-left
        conv_igemm(x, wpf, bias, out, k=(1, 2, 2), stride=1, pad=(0, 1, 1), cin=cin, cout=32, flags=flags | FO_DEPTH2SPACE,
                   mgrid=(Hi + 1, Wi + 1), oph=(1, cout))
        return
    conv_igemm(x, wpf, bias, out, k=(1, 3, 3), stride=1, pad=(0, 1, 1), cin=cin, cout=32, flags=flags | FO_DEPTH2SPACE,
               mgrid=(Hi, Wi), oph=(0, cout))


def convT_phases(x, wp4, bias, out, *, cin, cout, flags=0, mask=None, add=None):
    """k4 s2 p1 transposed conv (or the dgrad of a k4 s2 p1 conv) as 4 sub-pixel launches."""
    N, Hi, Wi, _ = x.shape
    per_phase = pad_out(cout) * 4 * cin
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        conv_igemm(x, wp4[ph * per_phase:(ph + 1) * per_phase], bias, out, k=(1, 2, 2), stride=1,
                   pad=(0, 1 - py, 1 - px), cin=cin, cout=cout, flags=flags, mask=mask, add=add,
                   ostride=2, oph=(py, px), mgrid=(Hi, Wi))


_ws_cache = {}


def _workspace(nbytes, device):
    """Split-K / column-sum scratch, one buffer PER STREAM: a buffer is allocated while its stream is current, so it
    lives in that stream's allocator pool and every kernel that touches it is ordered on that stream.  (A single
    shared buffer was a race: when it grew, the old one -- possibly still read by a reduce kernel on the side
    stream -- went back to the main stream's pool and was handed to the next activation.)"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        grow = 0 if buf is None else buf.numel() * 4 * 3 // 2
        buf = torch.empty((max(nbytes, grow) + 3) // 4 + 1024, device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    return buf


def conv_wgrad(P, Q, dw, dbias, *, T=1, k=(1, 3, 3), stride=1, pad=(0, 1, 1), a_real, b_real, in_relu=False):
    """dW[a][b][tap] = sum_m P[m][a] Q[qpix(m,tap)][b]; dbias (optional) = colsum(P)."""
    N, Hm, Wm, Ca = P.shape
    _, Hq, Wq, Cb = Q.shape
    d = _desc(N=N, T=T, Hin=Hq, Win=Wq, Hm=Hm, Wm=Wm, Hout=Hm, Wout=Wm, Cin=Cb, Cout=Ca,
              KD=k[0], KH=k[1], KW=k[2], stride=stride, padD=pad[0], padH=pad[1], padW=pad[2],
              ostride=1, ophH=0, ophW=0, ldIn=ld_of(Q), ldOut=ld_of(P), ldMask=0, ldAdd=0,
              flags=FO_IN_RELU if in_relu else 0)
    nbytes = _lib.load().fo_wgrad_ws_bytes(C.byref(d))
    if nbytes < 0:
        _lib.check(-1, "fo_wgrad_ws_bytes")
    ws = _workspace(nbytes, P.device)
    prof = PROFILER
    if prof is not None:
        wname = "conv_wgrad_%dx%d" % (Ca, Cb)
        if prof.detail:
            wname += f" [{N}x{Hm}x{Wm} k{k[0]}{k[1]}{k[2]} s{stride}]"
        nominal = 2.0 * N * Hm * Wm * Ca * Cb * k[0] * k[1] * k[2]
        prof.begin(wname, nominal * (temporal_share(T, k[0], pad[0]) if k[0] > 1 else 1.0), nominal)
    _lib.call("fo_conv_wgrad", C.byref(d), _ptr(P), _ptr(Q), _ptr(dw), a_real, b_real, _ptr(dbias), _ptr(ws),
              C.c_int64(ws.numel() * 4), _stream())
    if prof is not None:
        prof.end()


def bias_grad(g, dbias, c_real):
    rows = g.numel() // g.shape[-1] if g.is_contiguous() else g.shape[0] * g.shape[1] * g.shape[2]
    Cc = g.shape[-1]
    ws = _workspace(4 * Cc * 1024, g.device)
    _lib.call("fo_bias_grad", _ptr(g), _ptr(dbias), C.c_int64(rows), Cc, c_real, ld_of(g), _ptr(ws), _stream())


# ------------------------------------------------------------------ layout
def nchw_to_nhwc(x, cpad=None):
    N, Cc, H, W = x.shape
    cpad = cpad or Cc
    y = torch.empty((N, H, W, cpad), device=x.device, dtype=torch.float32)
    _lib.call("fo_nchw_to_nhwc", _ptr(dense_f32(x, "nchw_to_nhwc input")), _ptr(y), N, Cc, H, W, cpad, cpad, _stream())
    return y


def cat_nchw_to_nhwc8(a, b):
    """torch.cat([a, b], dim=1) of two NCHW tensors, written straight into the 8-channel NHWC input layout."""
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    assert b.shape == (N, Cb, H, W) and Ca + Cb <= 8
    y = torch.empty((N, H, W, 8), device=a.device, dtype=torch.float32)
    _lib.call("fo_nchw2_to_nhwc8", _ptr(dense_f32(a, "source frames")), Ca, _ptr(dense_f32(b, "background frames")), Cb, _ptr(y), N, H, W,
              _stream())
    return y


def nhwc_to_nchw(x, c_real, out=None, accumulate=False):
    N, H, W, _ = x.shape
    if out is None:
        out = torch.empty((N, c_real, H, W), device=x.device, dtype=torch.float32)
    _lib.call("fo_nhwc_to_nchw", _ptr(x), _ptr(out), N, c_real, H, W, ld_of(x), int(accumulate), _stream())
    return out


# ------------------------------------------------------------------ VQ
def vq_prepare(embed):
    embedT = torch.empty((512, 64), device=embed.device, dtype=torch.float32)
    enorm = torch.empty(512, device=embed.device, dtype=torch.float32)
    _lib.call("fo_vq_prepare", _ptr(embed), _ptr(embedT), _ptr(enorm), _stream())
    return embedT, enorm


def vq_assign(x, embedT, enorm, q_out, stats, train, stats_stream=None, force_ind=None):
    """x, q_out: [..., 64] views; stats: float32[1 + 512 + 512*64] = (sq_sum, counts, esum[512][64]); stats[0] is
    overwritten (an ordered sum of per-workgroup partials: bit-reproducible).  With train=True the EMA statistics (counts, esum)
    are written as well.  force_ind: teacher-forced codes (the search is skipped; see fo_vq_assign2)."""
    nvec = x.numel() // 64 if x.is_contiguous() else x.shape[0] * x.shape[1] * x.shape[2]
    ind = torch.empty(x.shape[:-1], device=x.device, dtype=torch.int64)
    f = _forced(force_ind, ind.shape, x.device)
    ws = _workspace(4096, x.device)
    if f is None:
        if LEDGER is not None:
            LEDGER.begin("vq_assign", 2.0 * nvec * 64 * 512)
        _lib.call("fo_vq_assign", _ptr(x), ld_of(x), C.c_int64(nvec), _ptr(embedT), _ptr(enorm), _ptr(ind), _ptr(q_out),
                  ld_of(q_out), _ptr(stats[0:1]), _ptr(ws), _stream())
    else:
        _lib.call("fo_vq_assign2", _ptr(x), ld_of(x), C.c_int64(nvec), _ptr(embedT), _ptr(enorm), _ptr(ind), _ptr(q_out), ld_of(q_out),
                  _ptr(stats[0:1]), None, 0, _ptr(f), _ptr(ws), _stream())
    if train:
        _vq_stats(x, nvec, ind, stats, stats_stream)
    return ind


def vq_ema(embed, cluster_size, embed_avg, stats, decay=0.99, eps=1e-5):
    _lib.call("fo_vq_ema", _ptr(embed), _ptr(cluster_size), _ptr(embed_avg), _ptr(stats[1:513]), _ptr(stats[513:]),
              C.c_float(decay), C.c_float(1 - decay), C.c_float(eps), _stream())


def vq_bwd(gq, x, q, gdiff, gx):
    nvec = x.shape[0] * x.shape[1] * x.shape[2]
    _lib.call("fo_vq_bwd", _ptr(gq), ld_of(gq), _ptr(x), ld_of(x), _ptr(q), ld_of(q), _ptr(gdiff),
              C.c_float(2.0 / (nvec * 64)), _ptr(gx), ld_of(gx), C.c_int64(nvec), _stream())


def vq_gather(ind, embedT, q_out):
    nvec = ind.numel()
    _lib.call("fo_vq_gather", _ptr(ind.contiguous()), _ptr(embedT), _ptr(q_out), ld_of(q_out), C.c_int64(nvec), _stream())


# ------------------------------------------------------------------ losses / optimiser
LOSS_WS_BYTES = 16384          # FO_LOSS_WS_BYTES (include/faceoff_hip.h): one partial per workgroup, added in order by a finish launch


def mse_slice_fwd(dec, gt_nchw, acc):
    """acc[0] = the sum of squares (overwritten; an ordered sum of per-workgroup partials: bit-reproducible)"""
    N, H, W, _ = dec.shape
    gt_nchw = dense_f32(gt_nchw, "ground truth")
    assert gt_nchw.shape[0] == N and gt_nchw.shape[2:] == (H, W), "ground truth must be [N,C,H,W] like the decoder output"
    _lib.call("fo_mse_slice_fwd", _ptr(dec), ld_of(dec), _ptr(gt_nchw), N, H, W, gt_nchw.shape[1], _ptr(acc), _ptr(_workspace(LOSS_WS_BYTES, dec.device)),
              _stream())


def mse_slice_bwd(dec, gt_nchw, gscale, gdec):
    N, H, W, _ = dec.shape
    gt_nchw = dense_f32(gt_nchw, "ground truth")
    c3 = gt_nchw.shape[1]
    _lib.call("fo_mse_slice_bwd", _ptr(dec), ld_of(dec), _ptr(gt_nchw), N, H, W, c3, _ptr(gscale),
              C.c_float(1.0 / (N * c3 * H * W)), _ptr(gdec), ld_of(gdec), _stream())


def mse_slice_fwd_bwd(dec, gt_nchw, acc, gscale, gdec):
    """sum of squares into acc (overwritten) AND the gradient into gdec, one pass (training step)"""
    N, H, W, _ = dec.shape
    gt_nchw = dense_f32(gt_nchw, "ground truth")
    assert gt_nchw.shape[0] == N and gt_nchw.shape[2:] == (H, W), "ground truth must be [N,C,H,W] like the decoder output"
    c3 = gt_nchw.shape[1]
    _lib.call("fo_mse_slice_fwd_bwd", _ptr(dec), ld_of(dec), _ptr(gt_nchw), N, H, W, c3, _ptr(gscale), C.c_float(1.0 / (N * c3 * H * W)),
              _ptr(gdec), ld_of(gdec), _ptr(acc), _ptr(_workspace(LOSS_WS_BYTES, dec.device)), _stream())


def adam_flat(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    b1, b2 = betas
    _lib.call("fo_adam_flat", _ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_int64(p.numel()), C.c_float(lr), C.c_float(b1),
              C.c_float(b2), C.c_float(eps), C.c_float(1 - b1 ** step), C.c_float(1 - b2 ** step), C.c_float(grad_scale),
              _stream())
    # the launch wrote p, m and v through raw pointers: tell torch's version counters (the engines skip repacking their filters while the
    # parameter arena's counter stands still -- VQVAEEngine.pack_filters, DiscEngine.pack_filters)
    for t in (p, m, v):
        torch.autograd.graph.increment_version(t)


def zero_(t):
    _lib.call("fo_zero", _ptr(t), C.c_int64(t.numel()), _stream())
    return t


def relu(x, out):
    rows = x.shape[0] * x.shape[1] * x.shape[2]
    _lib.call("fo_relu", _ptr(x), ld_of(x), _ptr(out), ld_of(out), C.c_int64(rows), x.shape[-1], _stream())
    return out


def add(a, b, out):
    rows = a.shape[0] * a.shape[1] * a.shape[2]
    _lib.call("fo_add", _ptr(a), ld_of(a), _ptr(b), ld_of(b), _ptr(out), ld_of(out), C.c_int64(rows), a.shape[-1], _stream())
    return out
