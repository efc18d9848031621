"""LPIPS / VGG-16 perceptual loss on the gfx950 kernels (reference models/lpips.py:51-161, loss.py:27-33).

`LPIPSEngine` holds the frozen VGG-16 feature weights (torchvision layout `features[0:30]`, sliced as in
lpips.py:125-134) and the five `lin` layers, runs both branches forward (ground truth: taps only;
reconstruction: every activation kept), evaluates the five fused tap heads and back-propagates to the
reconstruction only -- all LPIPS parameters are frozen (lpips.py:63-64,135-137), so the backward is
13 dgrads + 4 pool backwards + 5 head backwards, no wgrad.

Weights: `state_dict` keys follow the reference (`net.sliceK.<idx>.weight/bias`, `linK.model.1.weight`).
The pretrained files (torchvision VGG-16, vgg.pth from heibox, lpips.py:12-22) are a network download and
are NOT bundled: load them with `load_state_dict`; tests use seeded random weights (parity unpinned for the
pretrained values).

Two arithmetic modes (`dtype`): "fp32" -- everything on the exact-fp32 MFMA conv kernel (BASELINE config 2
arithmetic, 1e-3 parity with the reference's fp32 CPU path); "bf16" -- BASELINE config 3: activations, their
gradients and the frozen filters are stored as bf16 and contracted on the bf16 MFMA with fp32 accumulation
(csrc/conv_bf16.hip, csrc/lpips_bf16.hip), head arithmetic in fp32; parity is against the oracle's
bf16-simulated LPIPS (same rounding points), and the deviation from the fp32 oracle is reported next to it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops

VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
SLICE_OF = {}
for _s, (_a, _b) in enumerate([(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)], start=1):
    for _i in range(_a, _b):
        SLICE_OF[_i] = _s
TAP_CONVS = (1, 3, 6, 9, 12)            # conv ordinals whose ReLU output is relu1_2 ... relu5_3
SHIFT = (-.030, -.088, -.188)           # lpips.py:99
SCALE = (.458, .448, .450)              # lpips.py:100
_F3 = C.c_float * 3


LATE_HEADS_DEFAULT = "1"      # measured (tools/ab_env.sh 3, round 6, one box): config 3 33.80 / 33.84 / 33.85 -> 33.45 / 33.70 / 33.26 ms


def conv_keys():
    """[(state_dict prefix, Cin, Cout, pool_before)] for the 13 convs."""
    out, idx, cin, pool = [], 0, 3, False
    for v in VGG_CFG:
        if v == "M":
            idx += 1
            pool = True
            continue
        out.append((f"net.slice{SLICE_OF[idx]}.{idx}", cin, v, pool))
        cin, pool = v, False
        idx += 2
    return out


class LPIPSEngine:
    def __init__(self, state_dict, device, dtype="fp32"):
        assert dtype in ("fp32", "bf16")
        self.device = torch.device(device)
        self.bf16 = dtype == "bf16"
        self.act_dtype = torch.bfloat16 if self.bf16 else torch.float32
        self.convs = conv_keys()
        sd = {k: torch.as_tensor(v, dtype=torch.float32).to(self.device) for k, v in state_dict.items()}
        self.w, self.b, self.wp, self.wpd = [], [], [], []
        for i, (key, ci, co, _) in enumerate(self.convs):
            w, b = sd[key + ".weight"].contiguous(), sd[key + ".bias"].contiguous()
            self.w.append(w)
            self.b.append(b)
            if self.bf16:   # RGB layer: 3 -> 8 channels (one 16-byte pixel per tap), 9 -> 16 taps so that K = 128
                if i == 0:
                    wpad = torch.zeros((co, 8, 3, 3), device=self.device)
                    wpad[:, :3] = w
                    self.wp.append(ops.pack_conv_bf16(wpad, taps_pad=16))
                else:
                    self.wp.append(ops.pack_conv_bf16(w))
                self.wpd.append(ops.pack_conv_dgrad_bf16(w.reshape(co, ci, 9)))
                continue
            if i == 0:   # Cin 3 -> 8 channels, KW 3 -> 4 taps (zeros) so that K = 3*4*8 = 96 is a multiple of 32
                wpad = torch.zeros((co, 8, 3, 4), device=self.device)
                wpad[:, :3, :, :3] = w
                self.wp.append(ops.pack_conv(wpad))
            else:
                self.wp.append(ops.pack_conv(w))
            self.wpd.append(ops.pack_conv_dgrad(w.reshape(co, ci, 9)))     # frozen weights: packed once
        self.lin = [sd[f"lin{k}.model.1.weight"].reshape(-1).contiguous() for k in range(5)]
        self.shift, self.scale = _F3(*SHIFT), _F3(*SCALE)
        self.window_bytes = (1 << 31) - 1     # the conv kernel's buffer-descriptor window
        import os as _os
        # conv1_1 + conv1_2 in one launch (fo_vgg_conv1_fused_bf16): OPT-IN (FACEOFF_VGG_FUSE=1).  Correct (tests/test_lpips_gpu.py) but it does not pay:
        # 1.19-1.43 ms against 1.40 ms for the two launches without the relu1_1 output, 1.6-1.7 ms with it (tools/probes/vgg1_kernel_time.py)
        self.fuse_conv1 = bool(_os.environ.get("FACEOFF_VGG_FUSE"))
        self.force_fuse_conv1 = _os.environ.get("FACEOFF_BF16_FORCE_HALO", "0") not in ("", "0")     # tests: at any size
        # the pools of the reconstruction branch record their arg-max (2 bits per element) and the pool backwards read that instead of the full-size
        # input: FACEOFF_LPIPS_POOL_IDX=0 switches it off (A/B, tests)
        self.pool_idx = _os.environ.get("FACEOFF_LPIPS_POOL_IDX", "1") != "0"
        # round 6: the heads of the four taps that feed a max-pool run in the BACKWARD, fused with that pool's backward (fo_lpips_tap_fwd_bwd_unpool_bf16:
        # head gradient + un-pooled gradient summed in fp32, rounded once) instead of on a side stream at forward time followed by a pool-backward
        # pass: 2 x the tap's size less traffic per tap (5 GB per config-3 step), four launches fewer -- but the heads then sit IN the chain instead of
        # beside its first matrix-bound convolutions: measured -0.36 ms on config 3.  FACEOFF_LPIPS_LATE_HEADS=0: the side-stream heads + pool backwards
        self.late_heads = _os.environ.get("FACEOFF_LPIPS_LATE_HEADS", LATE_HEADS_DEFAULT) != "0"
        # ... and its convolutions leave the SIGN of every activation a data gradient will mask by as a bit plane (FACEOFF_LPIPS_MASK_BITS=0: the
        # data gradients read the bf16 activations themselves)
        self.mask_bits = _os.environ.get("FACEOFF_LPIPS_MASK_BITS", "1") != "0"

    # ------------------------------------------------------------------ pieces
    def _prep(self, src, nhwc):
        if nhwc:
            N, H, W, _ = src.shape
            ld = ops.ld_of(src)
        else:
            N, _, H, W = src.shape
            ld = 0
            src = src.contiguous()
        y = torch.empty((N, H, W, 8), device=self.device, dtype=self.act_dtype)
        _lib.call("fo_lpips_prep_bf16" if self.bf16 else "fo_lpips_prep", ops._ptr(src), int(nhwc), ld, ops._ptr(y), N, H, W,
                  self.shift, self.scale, ops._stream())
        return y

    def _conv(self, i, x, pooled=None, pool_idx=None, out_bits=None, pooled_bits=None):
        _, ci, co, _ = self.convs[i]
        N, H, W, _ = x.shape
        y = torch.empty((N, H, W, co), device=self.device, dtype=self.act_dtype)
        if self.bf16:
            ops.conv_bf16(x, self.wp[i], self.b[i], y, cin=8 if i == 0 else ci, cout=co, flags=ops.FO_OUT_RELU, pooled=pooled, pool_idx=pool_idx,
                          out_bits=out_bits, pooled_bits=pooled_bits)
        elif i == 0:
            ops.conv_igemm(x, self.wp[0], self.b[0], y, k=(1, 3, 4), pad=(0, 1, 1), cin=8, cout=co, flags=ops.FO_OUT_RELU)
        else:
            ops.conv_igemm(x, self.wp[i], self.b[i], y, k=(1, 3, 3), pad=(0, 1, 1), cin=ci, cout=co, flags=ops.FO_OUT_RELU)
        return y

    def _pool(self, x, want_idx=False, want_bits=False):
        """MaxPool2d(2); with want_idx (bf16 branch) also its arg-max codes, 2 bits per element (the backward then needs no look at x), with
        want_bits the ReLU-mask bit plane of the pooled tensor.  Returns (y, idx, bits)."""
        N, H, W, Cc = x.shape
        y = torch.empty((N, H // 2, W // 2, Cc), device=self.device, dtype=self.act_dtype)
        if want_idx and self.bf16:
            idx = torch.empty((N, H // 2, W // 2, Cc // 4), device=self.device, dtype=torch.uint8)
            bits = torch.empty((N, H // 2, W // 2, Cc // 8), device=self.device, dtype=torch.uint8) if want_bits else None
            _lib.call("fo_maxpool2_fwd_idx_bf16", ops._ptr(x), ops._ptr(y), ops._ptr(idx), ops._ptr(bits), N, H, W, Cc, ops._stream())
            return y, idx, bits
        _lib.call("fo_maxpool2_fwd_bf16" if self.bf16 else "fo_maxpool2_fwd", ops._ptr(x), ops._ptr(y), N, H, W, Cc, ops._stream())
        return y, None, None

    def features(self, x8, keep_all):
        """vgg16.forward (lpips.py:139-152).  Returns (taps[5], acts) where acts[i] = ReLU output of conv i,
        acts['p<i>'] = pooled input of conv i, acts['c<i>'] = that pool's arg-max codes or None, acts['b<i>'] / acts['pb<i>'] = the ReLU-mask
        bit planes of acts[i] / acts['p<i>'] or None (only when keep_all)."""
        taps, acts, x = [], {}, x8
        use_idx = self.bf16 and keep_all and self.pool_idx
        use_bits = use_idx and self.mask_bits             # (the planes replace the masks of the backward that also uses the codes)
        nxt = nxt_idx = nxt_bits = None                  # the pooled input of the next conv (its codes, its plane), when the previous launch already wrote it
        skip = 0
        if self.bf16 and self.fuse_conv1:
            # conv1_1 + conv1_2 (+ the pool in front of conv2_1) in ONE launch where frames are whole 4 x 32 tiles and the launch fills the chip:
            # relu1_1 is made in LDS, tile by tile, and only written out when the backward will need it as a ReLU mask (keep_all)
            N, H, W, _ = x8.shape
            if H % 4 == 0 and W % 32 == 0 and (N * (H // 4) * (W // 32) >= 4 * _lib.cu_count() or self.force_fuse_conv1) and N * H * W * 128 < (1 << 31):
                a0 = torch.empty((N, H, W, 64), device=self.device, dtype=self.act_dtype) if keep_all else None
                a1 = torch.empty((N, H, W, 64), device=self.device, dtype=self.act_dtype)
                nxt = torch.empty((N, H // 2, W // 2, 64), device=self.device, dtype=self.act_dtype)
                _lib.call("fo_vgg_conv1_fused_bf16", ops._ptr(x8), ops._ptr(self.wp[0]), ops._ptr(self.b[0]), ops._ptr(self.wp[1]), ops._ptr(self.b[1]),
                          ops._ptr(a0), ops._ptr(a1), ops._ptr(nxt), N, H, W, ops._stream())
                if keep_all:
                    acts[0], acts[1] = a0, a1
                taps.append(a1)
                x, skip = a1, 2
        for i, (_, ci, co, pool) in enumerate(self.convs):
            if i < skip:
                continue
            if pool:
                x, cidx, pbits = (nxt, nxt_idx, nxt_bits) if nxt is not None else self._pool(x, use_idx, use_bits)
                if keep_all:
                    acts[f"p{i}"], acts[f"c{i}"], acts[f"pb{i}"] = x, cidx, pbits
            nxt = nxt_idx = nxt_bits = None
            N, H, W, _ = x.shape
            # the max-pool in front of the NEXT conv rides along in this launch where the halo-tile kernel takes it (conv1_2): the pool's own
            # pass would read the full-resolution tap again
            if (self.bf16 and i + 1 < len(self.convs) and self.convs[i + 1][3] and i > 0 and ops.conv_bf16_pool_ok(N, H, W, ci, co)):
                nxt = torch.empty((N, H // 2, W // 2, co), device=self.device, dtype=self.act_dtype)
                if use_idx:
                    nxt_idx = torch.empty((N, H // 2, W // 2, co // 4), device=self.device, dtype=torch.uint8)
                if use_bits:
                    nxt_bits = torch.empty((N, H // 2, W // 2, co // 8), device=self.device, dtype=torch.uint8)
            # this activation's sign plane, where a data gradient will ask for it: conv i + 1 follows without a pool (a tap in front of a pool is
            # only ever looked at through the pool's codes)
            obits = None
            if use_bits and i + 1 < len(self.convs) and not self.convs[i + 1][3]:
                obits = torch.empty((N, H, W, co // 8), device=self.device, dtype=torch.uint8)
            x = self._conv(i, x, pooled=nxt, pool_idx=nxt_idx, out_bits=obits, pooled_bits=nxt_bits)
            if keep_all:
                acts[i], acts[f"b{i}"] = x, obits
            if i in TAP_CONVS:
                taps.append(x)
        return taps, acts

    # ------------------------------------------------------------------ loss (+ gradient into g_dec)
    def max_frames(self, H, W):
        """Frames per pass that keep relu1_2 (64 channels at full resolution) inside the conv kernel's 2 GiB window."""
        return max(1, self.window_bytes // (H * W * 64 * (2 if self.bf16 else 4)))

    def target_taps(self, gt_nchw):
        """The five taps of the ground-truth branch (no dependence on the model: the trainer computes them on a side
        stream while the VQ-VAE forward runs).  None when the batch needs frame chunking."""
        N, _, H, W = gt_nchw.shape
        if N > self.max_frames(H, W):
            return None
        taps0, _ = self.features(self._prep(gt_nchw, nhwc=False), keep_all=False)
        return taps0

    def loss_and_grad(self, gt_nchw, dec_nhwc, g_dec=None, weight=1.0, gscale=None, taps0=None):
        """perceptual = LPIPS(gt, dec[..., :3]).mean() (loss.py:33).  If g_dec (NHWC, same pixel stride as dec)
        is given, adds weight * gscale * d perceptual / d dec[..., :3] to it.  Returns the loss as a [1] tensor.
        Frames are independent in LPIPS, so large batches run in frame chunks that keep every activation
        inside the conv kernel's 2 GiB buffer-descriptor window (relu1_2 is 64 channels at full resolution)."""
        N, H, W, _ = dec_nhwc.shape
        max_frames = self.max_frames(H, W)
        if N > max_frames:
            nchunks = -(-N // max_frames)
            per = -(-N // nchunks)
            if gscale is None:
                gscale = torch.ones(1, device=self.device)
            total = torch.zeros(1, device=self.device)
            vals = []
            for a in range(0, N, per):
                b = min(N, a + per)
                frac = (b - a) / N
                part = self._loss_and_grad(gt_nchw[a:b], dec_nhwc[a:b], None if g_dec is None else g_dec[a:b], weight,
                                           gscale * frac)
                total += part * frac
                vals.append(self.last_per_image)
            self.last_per_image = torch.cat(vals)
            return total
        return self._loss_and_grad(gt_nchw, dec_nhwc, g_dec, weight, gscale, taps0)

    def _loss_and_grad(self, gt_nchw, dec_nhwc, g_dec, weight, gscale, taps0=None):
        x1 = self._prep(dec_nhwc, nhwc=True)
        N, H, W, _ = x1.shape
        if taps0 is None:
            taps0, _ = self.features(self._prep(gt_nchw, nhwc=False), keep_all=False)
        taps1, acts = self.features(x1, keep_all=g_dec is not None)
        fused = self.bf16 and g_dec is not None          # training on the bf16 branch: value and gradient of a tap from ONE pass over its two maps
        if gscale is None:
            gscale = torch.ones(1, device=self.device)
        head, head_ready = [], [None] * 5
        late = False
        if fused:
            # The backward chain starts at the deepest tap and meets the shallower taps' gradients only in the pool backwards, much later: only
            # tap 5's head runs in the chain's way.  The other four -- HBM-bound passes over the big feature maps, 1.5 ms at C3 -- run on a side
            # stream beside the first (matrix-bound) data-gradient convolutions.  Each head adds into its own row of `vals`; the rows are summed
            # in a fixed order afterwards, so the loss does not depend on how the streams interleave.
            import os as _os
            main = torch.cuda.current_stream(self.device)
            # (head_overlap: the trainer switches it off together with the engine's side streams -- bench.py's per-kernel region and
            # `--serial-streams` traces want every launch alone on the GPU; round 4's profiles had conv5_x's data gradients stretched 2.5-7x
            # by the heads running beside them, which the serialised --pmc passes did not show)
            overlap = getattr(self, "head_overlap", True) and not _os.environ.get("FACEOFF_NO_LPIPS_HEAD_OVERLAP")
            if overlap and getattr(self, "_head_stream", None) is None:
                self._head_stream = torch.cuda.Stream(device=self.device)
            vals = torch.zeros((5, N), device=self.device)
            late = self.late_heads and self.pool_idx and all(acts.get(f"c{i}") is not None for i, cv in enumerate(self.convs) if cv[3])
            head = [torch.empty_like(t) if (k == 4 or not late) else None for k, t in enumerate(taps1)]

            def run_head(k):
                n, h, w, c = taps1[k].shape
                ws = ops._workspace(_lib.load().fo_lpips_tap_ws_bytes_bf16(n, h, w, c), self.device)      # (per stream: the heads run on two)
                _lib.call("fo_lpips_tap_fwd_bwd_bf16", ops._ptr(taps0[k]), ops._ptr(taps1[k]), ops._ptr(self.lin[k]), ops._ptr(vals[k]), ops._ptr(gscale),
                          ops._ptr(head[k]), n, h, w, c, ops._ptr(ws), ops._stream())
            run_head(4)
            if late:
                pass                                     # taps 1-4: in the backward, with their pools' backward (below)
            elif overlap:
                side = self._head_stream
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    for k in (3, 2, 1, 0):               # in the order the chain will ask for them
                        run_head(k)
                        head_ready[k] = torch.cuda.Event()
                        head_ready[k].record(side)
                # (no record_stream on these tensors -- up to 1.3 GB each: it would keep the allocator from handing their blocks to the next
                # step until the side stream's work has been observed complete, and a host that runs a few steps ahead then grows the pool by
                # gigabytes in the middle of training.  It is not needed: every head is awaited by a pool backward on the main stream below,
                # so by the time this function's tensors are freed, in main-stream order, the side stream is done with them.)
            else:
                for k in (3, 2, 1, 0):
                    run_head(k)
            val = None
        else:
            val = torch.zeros(N, device=self.device)
            for k in range(5):
                n, h, w, c = taps1[k].shape
                nb = _lib.load().fo_lpips_tap_ws_bytes_bf16(n, h, w, c) if self.bf16 else _lib.load().fo_lpips_tap_ws_bytes(n, h, w)
                _lib.call("fo_lpips_tap_fwd_bf16" if self.bf16 else "fo_lpips_tap_fwd", ops._ptr(taps0[k]), ops._ptr(taps1[k]), ops._ptr(self.lin[k]),
                          ops._ptr(val), n, h, w, c, ops._ptr(ops._workspace(nb, self.device)), ops._stream())
            loss = val.mean().reshape(1)
            self.last_per_image = val
        if g_dec is None:
            return loss
        # ---- backward, deepest stage first
        for k in range(5 if not fused else 0):
            n, h, w, c = taps1[k].shape
            g = torch.empty_like(taps1[k])
            _lib.call("fo_lpips_tap_bwd_bf16" if self.bf16 else "fo_lpips_tap_bwd", ops._ptr(taps0[k]), ops._ptr(taps1[k]), ops._ptr(self.lin[k]), ops._ptr(gscale),
                      ops._ptr(g), n, h, w, c, ops._stream())
            head.append(g)
        g = head[4]                                     # grad wrt (pre-ReLU) output of conv 12
        for i in range(12, -1, -1):
            _, ci, co, pool = self.convs[i]
            if i == 0:
                gin = torch.zeros((N, H, W, 8), device=self.device, dtype=self.act_dtype)
                self._dgrad(g, 0, gin, co, 3, None)
                g = gin
                break
            if pool:                                     # conv i reads the pooled tensor
                gp = torch.empty_like(acts[f"p{i}"])
                cidx = acts.get(f"c{i}")
                # with the pool's arg-max codes the pool backward never looks at its input: the one ReLU mask it applied to the pooled-through
                # gradient (max > 0) moves into this data gradient's epilogue as mask = the pooled tensor, a quarter of the input's size
                self._dgrad(g, i, gp, co, ci, acts[f"p{i}"] if cidx is not None else None, acts.get(f"pb{i}"))
                x = acts[i - 1]                          # pre-pool tensor = ReLU output of conv i-1 = a LPIPS tap
                tap = TAP_CONVS.index(i - 1)
                gx = torch.empty_like(x)
                n, h, w, c = x.shape
                if head_ready[tap] is not None:
                    torch.cuda.current_stream(self.device).wait_event(head_ready[tap])
                if fused and late:                       # this tap's head and its pool's backward in one pass over the two feature maps
                    ws = ops._workspace(_lib.load().fo_lpips_tap_ws_bytes_bf16(n, h, w, c), self.device)
                    _lib.call("fo_lpips_tap_fwd_bwd_unpool_bf16", ops._ptr(taps0[tap]), ops._ptr(x), ops._ptr(self.lin[tap]), ops._ptr(vals[tap]),
                              ops._ptr(gscale), ops._ptr(gp), ops._ptr(cidx), ops._ptr(gx), n, h, w, c, ops._ptr(ws), ops._stream())
                elif cidx is not None:                   # (head[tap] is zero wherever x is: lpips_head_*_bf16 apply the tap's own ReLU mask)
                    _lib.call("fo_maxpool2_bwd_idx_bf16", ops._ptr(cidx), ops._ptr(gp), ops._ptr(head[tap]), ops._ptr(gx), n, h, w, c, ops._stream())
                else:
                    _lib.call("fo_maxpool2_bwd_bf16" if self.bf16 else "fo_maxpool2_bwd", ops._ptr(x), ops._ptr(gp), ops._ptr(head[tap]), ops._ptr(gx), n, h, w, c,
                              ops._stream())
                g = gx
            else:
                gin = torch.empty_like(acts[i - 1])
                self._dgrad(g, i, gin, co, ci, acts[i - 1], acts.get(f"b{i - 1}"))
                g = gin
        one = torch.ones(1, device=self.device)          # gscale already went into the tap gradients
        if self.bf16:
            _lib.call("fo_lpips_prep_bwd_bf16", ops._ptr(g), ops._ptr(g_dec), ops.ld_of(g_dec), C.c_int64(N * H * W), self.scale,
                      ops._ptr(one), C.c_float(weight), ops._stream())
        else:
            _lib.call("fo_lpips_prep_bwd", ops._ptr(g), 8, ops._ptr(g_dec), ops.ld_of(g_dec), C.c_int64(N * H * W), self.scale,
                      ops._ptr(one), C.c_float(weight), ops._stream())
        if fused:
            val = vals.sum(0)                            # (every head has been awaited by a pool backward by now)
            loss = val.mean().reshape(1)
            self.last_per_image = val
        return loss

    def _dgrad(self, g, i, out, cin, cout, mask, mask_bits=None):
        """Data gradient of conv i: the same 3x3 contraction with the flipped / channel-swapped filter; the ReLU mask from its bit plane where
        the forward left one (1/16 of the activation's bytes)."""
        if self.bf16:
            if mask_bits is not None:
                ops.conv_bf16(g, self.wpd[i], None, out, cin=cin, cout=cout, mask_bits=mask_bits)
            else:
                ops.conv_bf16(g, self.wpd[i], None, out, cin=cin, cout=cout, mask=mask)
        else:
            ops.conv_igemm(g, self.wpd[i], None, out, k=(1, 3, 3), pad=(0, 1, 1), cin=cin, cout=cout, mask=mask)
