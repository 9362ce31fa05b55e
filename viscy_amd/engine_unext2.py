"""Kernel schedule of UNeXt2 forward / backward on MI355X (see viscy_amd/unext2.py).

``Engine`` owns
  * the flat fp32 master-parameter buffer (every nn.Parameter of the model is a view into it, in
    *reverse forward order* so gradient buckets become ready front-to-back during backward) and
    the matching flat gradient buffer;
  * ``prepare()``: per-step re-layout of master weights into GEMM operands (compute dtype);
  * ``forward()`` / ``backward()``: the explicit launch sequences.

``ops`` is the kernel backend: ``viscy_amd.ops`` (HIP, the only backend the product ever uses).
Tests may inject a reference implementation of the same call surface to validate the schedule
itself on CPU (tests/ref_ops.py); nothing in the package does.
"""

from __future__ import annotations

import os

import torch
from torch import Tensor

from . import _lib as L

CHUNK_MB_DEFAULT = "0"  # see Engine._sample_chunk


class _BlockW:
    """prepared operands + parameter handles of one ConvNeXt-V2 block"""

    __slots__ = ("C", "p", "dw_w", "W1f", "W1fT", "b1f", "W2", "W2T", "v1", "fc2_w", "fc2_b", "grn_w", "grn_b", "dp", "img", "img2", "b2f")


class _ProjW:
    __slots__ = ("cin", "cout", "taps", "ln", "conv", "W", "WT")


def _views(flat: Tensor, shapes):
    out, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= s
        out.append(flat[off : off + n].view(shp))
        off += n
    return out


class _ZeroArena:
    """One zero-filled fp32 allocation per pass, handed out as views: the ~130 small reduction targets of a step
    (GRN sums, bias / weight gradient staging, InstanceNorm sums) used to cost one fill launch each (154 launches,
    0.74 ms per step in the kernel trace).  Sized from the previous pass with the same key; a pass that needs more than
    was recorded falls back to individual allocations (and records the new size)."""

    def __init__(self, dev, numel: int, ops=None):
        self.dev, self.ops = dev, ops
        self.buf = self._zeros(numel) if numel > 0 else None
        self.off = 0
        self.used = 0

    def _zeros(self, *shape) -> Tensor:
        if self.ops is not None and hasattr(self.ops, "zeros"):
            return self.ops.zeros(*shape, device=self.dev)  # vsx_fill_f32: the captured step holds no ATen fill
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def take(self, *shape) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) // 64 * 64  # 256-byte granules: every view stays 16-byte aligned
        self.used += n_al
        if self.buf is not None and self.off + n_al <= self.buf.numel():
            v = self.buf[self.off:self.off + n].view(*shape)
            self.off += n_al
            return v
        return self._zeros(*shape)


class Engine:
    def __init__(self, model, ops=None):
        if ops is None:
            from . import ops as hip_ops

            ops = hip_ops
        self.ops = ops
        self.model = model
        self.cfg = model.cfg
        self._zeros4c = {}       # C -> zeros(4C): the identity GRN of ConvNeXt-V1 blocks
        self._dp_inject = None   # tests: list of per-sample branch scales [B], consumed in forward block order
        self._za = None          # zero arena of the pass in flight
        self._za_need = {}       # (pass, B, H, W) -> fp32 elements the pass took last time
        params = list(model.parameters())
        self.device = params[0].device
        # ---- flat parameter / gradient buffers, reverse forward order (head first, stem last)
        order = self._param_order()
        assert len(order) == len(params) and len({id(p) for p in order}) == len(order)
        total = sum(p.numel() for p in order)
        # keep every slice 16-byte aligned
        offs, off = [], 0
        for p in order:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad_of = {}
        with torch.no_grad():
            for p, o in zip(order, offs):
                v = self.flat[o : o + p.numel()].view(p.shape)
                v.copy_(p.detach().to(torch.float32))
                p.data = v
                self.grad_of[id(p)] = self.flat_grad[o : o + p.numel()].view(p.shape)
        self.order, self.offsets, self.numel = order, offs, total
        self.bucket_bounds = self._bucket_bounds()
        self.on_bucket_ready = None  # callable(bucket_index) set by viscy_amd.parallel
        self._pending_bwd = 0        # forwards of the current step whose backward has not run yet
        self._prepared_for = None
        self.W = None

    # ------------------------------------------------------------------ parameter ordering
    def _stage_params_rev(self, stage):
        ps = []
        for blk in reversed(list(stage.blocks)):
            if hasattr(blk, "gamma"):  # ConvNeXt-V1 block: layer scale, no GRN
                ps += [blk.gamma, blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.norm.weight,
                       blk.norm.bias, blk.conv_dw.weight, blk.conv_dw.bias]
                continue
            ps += [blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.mlp.grn.weight, blk.mlp.grn.bias, blk.mlp.fc1.weight,
                   blk.mlp.fc1.bias, blk.norm.weight, blk.norm.bias, blk.conv_dw.weight, blk.conv_dw.bias]
        if not isinstance(stage.downsample, torch.nn.Identity):
            ps += [stage.downsample[1].weight, stage.downsample[1].bias, stage.downsample[0].weight,
                   stage.downsample[0].bias]
        return ps

    def _param_order(self):
        m = self.model
        ps = []
        if self.cfg.get("head", "conv") == "conv":
            ps = [m.head.conv[1].weight, m.head.conv[1].bias, m.head.conv[0].adn.A.weight, m.head.conv[0].conv.weight,
                  m.head.conv[0].conv.bias]
        self._bucket_marks = [0]
        if self.cfg.get("head") == "embed":  # ContrastiveEncoder: pooled-embedding tail instead of decoder + head
            t = m.tail
            ps = [t.bn4.weight, t.bn4.bias, t.fc3.weight, t.fc3.bias, t.bn1.weight, t.bn1.bias, t.fc0.weight, t.fc0.bias,
                  t.norm.weight, t.norm.bias]
        else:
            for st in reversed(list(m.decoder.decoder_stages)):
                ps += self._stage_params_rev(st.conv)
                if getattr(st, "pre_conv", None) is not None:  # decoder_upsample_pre_conv: runs before the stage in the forward
                    ps += [st.pre_conv.weight, st.pre_conv.bias]
        self._bucket_marks.append(len(ps))  # bucket 0 = head + decoder (or the embedding tail)
        for i in (3, 2):
            ps += self._stage_params_rev(getattr(m.encoder_stages, f"stages_{i}"))
        self._bucket_marks.append(len(ps))  # bucket 1 = encoder stages 3, 2
        for i in (1, 0):
            ps += self._stage_params_rev(getattr(m.encoder_stages, f"stages_{i}"))
        ps += [m.encoder_stages.stem_1.weight, m.encoder_stages.stem_1.bias, m.stem.conv.weight, m.stem.conv.bias]
        if getattr(m, "stem2d", None) is not None:  # FCMAE: the Conv2d stem of Z == 1 inputs
            ps += [m.stem2d.weight, m.stem2d.bias]
        self._bucket_marks.append(len(ps))  # bucket 2 = encoder stages 1, 0 + stem
        return ps

    def _bucket_bounds(self):
        b = []
        for i in range(len(self._bucket_marks) - 1):
            lo = self.offsets[self._bucket_marks[i]]
            hi_idx = self._bucket_marks[i + 1]
            hi = self.offsets[hi_idx] if hi_idx < len(self.offsets) else self.flat.numel()
            b.append((lo, hi))
        return b

    def g(self, p) -> Tensor:
        return self.grad_of[id(p)]

    def encoder_frozen(self) -> bool:
        """``model.encoder.requires_grad_(False)`` (cytoland engine.py:204-206, the FCMAE fine-tuning recipe): every
        parameter from the encoder bucket on (encoder stages + stem: the tail of the flat buffer) is frozen — the backward
        stops after the decoder and the optimiser skips the tail.  Any other pattern of frozen parameters is not built."""
        lo = self._bucket_marks[1]
        frozen = [not p.requires_grad for p in self.order]
        if not any(frozen):
            return False
        if all(frozen[lo:]) and not any(frozen[:lo]):
            return True
        raise NotImplementedError("only a fully frozen encoder (+ stem) with a fully trainable decoder / head is built")

    def trainable_numel(self) -> int:
        return self.offsets[self._bucket_marks[1]] if self.encoder_frozen() else self.flat.numel()

    def attach_grads(self) -> None:
        for p in self.order:
            p.grad = self.grad_of[id(p)]

    # ------------------------------------------------------------------ weight preparation
    def _prep_block(self, blk, dt, need_bwd):
        o = self.ops
        w = _BlockW()
        C = blk.conv_dw.weight.shape[0]
        w.C, w.p = C, blk
        w.dw_w = torch.empty((49, C), dtype=torch.float32, device=self.device)
        o.transpose_f32(blk.conv_dw.weight, w.dw_w, C, 49, False)
        w.W1f, w.W1fT = o.prep_weight(blk.mlp.fc1.weight, 4 * C, C, 1, dt, want=True, want_t=need_bwd,
                                      gamma=blk.norm.weight)
        w.b1f = o.matvec(blk.mlp.fc1.weight, blk.norm.bias, blk.mlp.fc1.bias, 4 * C, C)
        w.dp = 0.0  # stochastic-depth rate of this block (set by prepare() from cfg["drop_path"])
        w.img = None  # fragment-major LDS image of (W1f, W2) for the fused GRN-MLP kernel, packed on first use
        w.v1 = hasattr(blk, "gamma")
        if w.v1:
            # ConvNeXt-V1: y = x + gamma * fc2(gelu(fc1(.))) — the layer scale is folded into fc2 (vsx_layer_scale_fold) and
            # the block runs the V2 schedule with a zero (identity) GRN; the fc2 gradients are unfolded in _block_bwd
            w.fc2_w, w.fc2_b = o.layer_scale_fold(blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.gamma)
            z = self._zeros4c.get(C)
            if z is None:
                z = self._zeros4c[C] = torch.zeros(4 * C, dtype=torch.float32, device=self.device)
            w.grn_w = w.grn_b = z
        else:
            w.fc2_w, w.fc2_b = blk.mlp.fc2.weight, blk.mlp.fc2.bias
            w.grn_w, w.grn_b = blk.mlp.grn.weight, blk.mlp.grn.bias
        w.W2, w.W2T = o.prep_weight(w.fc2_w, C, 4 * C, 1, dt, want=True, want_t=need_bwd)
        # fc2 bias with the GRN beta folded in (b2 + W2 . beta): what fc2 needs when the GRN scale lives in per-sample weights
        w.b2f = o.matvec(w.fc2_w, w.grn_b, w.fc2_b, C, 4 * C) if dt == torch.bfloat16 and C > 64 else None
        w.img2 = None  # image of (W2^T, W2) for the block backward without a stored dz
        return w

    def _prep_proj(self, ln, conv, dt, need_bwd):
        o = self.ops
        w = _ProjW()
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        taps = conv.weight.shape[2] * conv.weight.shape[3]
        w.cin, w.cout, w.taps, w.ln, w.conv = cin, cout, taps, ln, conv
        w.W, w.WT = o.prep_weight(conv.weight, cout, cin, taps, dt, want=True, want_t=need_bwd)
        return w

    def prepare(self, dt: torch.dtype, need_bwd: bool):
        key = (dt, need_bwd)
        m, cfg, o = self.model, self.cfg, self.ops
        with o.batch():  # ~150 weight-space jobs -> a handful of task-list launches (ops.batch)
            W = self._prepare_ops(dt, need_bwd)
        if dt == torch.bfloat16 and self._mlp_flag() and hasattr(o, "mlp_pack"):
            # fragment-major weight images of the fused GRN-MLP kernels, every block's in one task list (their inputs are the
            # outputs of the list above); a block whose shape the fused kernels do not serve simply never reads its image
            bwd_img = need_bwd and bool(self._mlp_flag() & 8)
            with o.batch():
                for _, blocks in list(W["enc"]) + list(W["dec"]):
                    for w in blocks:
                        if any(o.mlp_supported(w.C, 1 << 20, 1 << 20, dt, md) for md in (None, 2, 4)):  # a width the kernels serve
                            w.img = o.mlp_pack(w.W1f, w.W2, w.C)
                            if bwd_img:
                                w.img2 = o.mlp_pack(w.W2T, w.W2, w.C)
        self.W = W
        self._prepared_for = key
        return W

    def _prepare_ops(self, dt: torch.dtype, need_bwd: bool):
        m, cfg, o = self.model, self.cfg, self.ops
        W = {"dt": dt}
        # stem: [Cout3d, K] (block-diagonal expansion when the stem keeps D' > 1 depth slabs)
        sw = m.stem.conv.weight
        co3, K = sw.shape[0], sw[0].numel()
        Dp = cfg["ratio"]
        if Dp == 1:
            W["stem_W"], _ = o.prep_weight(sw, co3, K, 1, dt)
            if dt == torch.bfloat16 and K % 32:
                o.flush()  # pad_cols reads what the task list has yet to write
                # K = 80 (the 5x4x4 single-channel stem) is not a whole number of 32-deep MFMA slabs: the projection fell to
                # the generic GEMM (2.9 ms at B = 512, 250 GB/s).  Zero-padded to 96 on both operands it runs on the lean one.
                W["stem_W"] = o.pad_cols(W["stem_W"], (K + 31) // 32 * 32)
            W["stem_b"] = m.stem.conv.bias
        else:  # rare configuration (Z = 15): tiny weight-space expansion done with tensor ops
            we = torch.zeros((co3, Dp, Dp, K), dtype=torch.float32, device=self.device)
            for d in range(Dp):
                we[:, d, d, :] = sw.detach().reshape(co3, K)
            W["stem_W"] = we.reshape(co3 * Dp, Dp * K).to(dt).contiguous()
            W["stem_b"] = m.stem.conv.bias.detach().repeat_interleave(Dp).contiguous()
        s2 = getattr(m, "stem2d", None)
        if s2 is not None:  # FCMAE: Conv2d stem of Z == 1 inputs (fcmae.py:348-353,369-370) — the same patch GEMM with kz = 1
            K2 = s2.weight[0].numel()
            W["stem2d_W"], _ = o.prep_weight(s2.weight, s2.weight.shape[0], K2, 1, dt)
            if dt == torch.bfloat16 and K2 % 32:
                o.flush()
                W["stem2d_W"] = o.pad_cols(W["stem2d_W"], (K2 + 31) // 32 * 32)
            W["stem2d_b"] = s2.bias
        enc = []
        for i in range(4):
            st = getattr(m.encoder_stages, f"stages_{i}")
            proj = None
            if not isinstance(st.downsample, torch.nn.Identity):
                proj = self._prep_proj(st.downsample[0], st.downsample[1], dt, need_bwd)
            enc.append((proj, [self._prep_block(b, dt, need_bwd) for b in st.blocks]))
        rates = cfg.get("drop_path")  # per encoder block, forward order (timm: linspace over all blocks; FCMAE: constant)
        if rates:
            flat_blocks = [bw for _, blocks in enc for bw in blocks]
            assert len(rates) == len(flat_blocks)
            for bw, r in zip(flat_blocks, rates):
                bw.dp = float(r)
        W["enc"] = enc
        dec = []
        if cfg.get("head") == "embed":
            t = m.tail
            E, Cf = t.fc0.weight.shape
            P = t.fc3.weight.shape[0]
            W["fc0"], W["fc0T"] = o.prep_weight(t.fc0.weight, E, Cf, 1, torch.float32, want=True, want_t=need_bwd)
            W["fc3"], W["fc3T"] = o.prep_weight(t.fc3.weight, P, E, 1, torch.float32, want=True, want_t=need_bwd)
        else:
            for us in m.decoder.decoder_stages:
                st = us.conv
                proj = self._prep_proj(st.downsample[0], st.downsample[1], dt, need_bwd)
                dec.append((proj, [self._prep_block(b, dt, need_bwd) for b in st.blocks]))
            # decoder_upsample_pre_conv: Conv2d(C, C, 3, padding 1) in front of each pixel shuffle = patch matrix x [C, 9C]
            pre = []
            for us in m.decoder.decoder_stages:
                pc = getattr(us, "pre_conv", None)
                if pc is None:
                    pre.append(None)
                else:
                    cpc = pc.weight.shape[0]
                    pre.append((pc, *o.prep_weight(pc.weight, cpc, cpc, 9, dt, want=True, want_t=need_bwd)))
            W["dec_pre"] = pre
        W["dec"] = dec
        if cfg.get("head", "conv") == "conv":
            hc = m.head.conv[0].conv
            cmid, c3 = hc.weight.shape[0], hc.weight.shape[1]
            hdt = self._head_dtype(dt)
            W["head_Wc"], _ = o.prep_weight(hc.weight, cmid, c3, 27, hdt, tapmode=1)
            if need_bwd:
                if hdt == torch.bfloat16 and c3 == 8 and cmid == 32 and cfg["out_stack_depth"] == 5:
                    o.flush()  # reads head_Wc
                    W["head_Wp"] = o.head_conv_dgrad_prep(W["head_Wc"])  # direct LDS-tiled dgrad (csrc/headconv.hip)
                W["head_Wd"] = o.prep_head_dgrad(hc.weight, cmid, c3, cfg["out_stack_depth"], hdt)
        return W

    def _head_dtype(self, dt: torch.dtype) -> torch.dtype:
        """PixelToVoxelHead works on 4 * out_channels channels per depth plane: with out_channels odd (1 -> 1 virtual staining)
        that is not a whole number of 16-byte bf16 vectors, and the (small) head then runs in fp32 behind a bf16 trunk"""
        if dt == torch.bfloat16 and self.model.head.conv[0].conv.weight.shape[1] % 8:
            return torch.float32
        return dt

    # ------------------------------------------------------------------ block forward / backward
    def _mlp_flag(self) -> int:
        return L.lib().vsx_get_flag(b"mlp_fused") if self.ops.__name__.endswith("viscy_amd.ops") else 0

    def _mlp_mode(self, C, hw, M, dt, training: bool) -> bool:
        """fused GRN-MLP kernel available for this block shape (bf16, C a supported width, whole workgroup tiles per sample)
        and enabled (``mlp_fused`` flag: bit 0 = inference forward (statistics + output passes), bit 1 = training fc1 (the
        statistics pass that also stores h and g; fc2 stays the unfused GEMM, which reads the stored g))"""
        if dt != torch.bfloat16:
            return False
        flag = self._mlp_flag()
        if training:
            return bool(flag & 2) and self.ops.mlp_supported(C, hw, M, dt, 2)
        return bool(flag & 1) and self.ops.mlp_supported(C, hw, M, dt)

    def _mlp_bwd_fused(self, C, hw, M, dt, B) -> int:
        """``mlp_fused`` bit 3: the block backward writes dh once (csrc/mlp.hip MODE 4) instead of dz and then dh.  Returns
        0 (unfused), 1 (GRN statistics and fc2 weight gradient from the per-sample products dout_b^T g_b) or 2 (statistics
        from the MODE 3 pass: when the B per-sample [C, 4C] fp32 products would be larger than 512 MB)"""
        if dt != torch.bfloat16 or not (self._mlp_flag() & 8) or not self.ops.mlp_supported(C, hw, M, dt, 4):
            return 0
        if B * C * 4 * C * 4 <= (512 << 20) and C >= 96 and hw % 64 == 0:
            return 1
        return 2 if self.ops.mlp_supported(C, hw, M, dt, 3) else 0

    def _mlp_recompute_h(self, C, hw, M, dt, B) -> bool:
        """``mlp_fused`` bit 6: the training fc1 stores g only and the dh pass recomputes h (needs the fused backward)"""
        return bool(self._mlp_flag() & 64) and self._mlp_bwd_fused(C, hw, M, dt, B) != 0 and self.ops.mlp_supported(C, hw, M, dt, 5) \
            and self.ops.mlp_supported(C, hw, M, dt, 6)

    def _mlp_drop_xh(self, C, hw, M, dt, B) -> bool:
        """``mlp_fused`` bit 7 (with 5 and 6): the normalised rows x^ are not stored either — the forward keeps the depthwise
        output y and the row mean / rstd; the dh pass re-normalises y and hands on dh * rstd (csrc/mlp.hip MODE 7), the fc1
        weight gradient is a plain TN GEMM on y with a rank-1 correction, the data-gradient GEMM's LayerNorm epilogue re-forms
        x^ from y.  Needs that epilogue (C <= 256): the stand-alone LayerNorm backward reads a stored x^."""
        flag = self._mlp_flag()
        if str(C) in os.environ.get("VSX_DROPXH_SKIP", "").split(","):  # A/B knob: widths that keep x^ stored (MODE 5) under bit 7
            return False
        return bool(flag & 128) and bool(flag & 32) and self._mlp_recompute_h(C, hw, M, dt, B) and self.ops.mlp_supported(C, hw, M, dt, 7) \
            and hasattr(self.ops, "dgrad_ln_bwd") and bool(L.lib().vsx_gemm_nt_ln_bwd_supported(M, C, 4 * C, L.dtype_code(dt)))

    def _sample_chunk(self, B, hw, C, which: str = "bwd") -> int:
        """samples per chunk of the sample-chunk-major schedule, 0 = whole batch in one launch.  ``VSX_CHUNK_MB`` = target size of
        the 4C-wide chunk in MiB (0 / unset: off); a chunk holds at least 256 row tiles of 256 rows (one per CU) and the
        schedule only applies when the batch has at least two chunks"""
        mb = float(os.environ.get("VSX_CHUNK_MB", CHUNK_MB_DEFAULT))
        if mb <= 0 or hw % 256 or os.environ.get("VSX_CHUNK_" + which.upper(), "1") == "0":
            return 0
        per_sample = hw * 4 * C * 2
        n = max(int(mb * (1 << 20)) // per_sample, 1)
        n = max(n, -(-256 * 256 // hw))
        return n if n * 2 <= B else 0

    def _block_fwd(self, x, w, B, H, Wd, dt, save, rows=None):
        """One ConvNeXt-V2 block on a dense channels-last map [B*H*W, C].  ``rows = (idx, inv, keep, L)`` selects the FCMAE
        masked path (fcmae.py:196-230): ``x`` arrives already multiplied by the mask, the depthwise convolution runs dense,
        LayerNorm / GRN-MLP run on the L kept tokens per sample only (compact rows), and the result is scattered back into
        zeros and added to the (masked) shortcut."""
        o = self.ops
        C, M = w.C, B * H * Wd
        blk = w.p
        hw = H * Wd
        y = o.dwconv7_fwd(x, w.dw_w, blk.conv_dw.bias, B, H, Wd, C)
        xres = x
        if rows is not None:
            idx, inv, _, Lr = rows
            M, hw = B * Lr, Lr
            y = o.rows_select(y, idx, M, C)      # masked_patchify
            xres = o.rows_select(x, idx, M, C)   # the shortcut of the kept tokens
        fused = self._mlp_mode(C, hw, M, dt, save is not None)
        # mlp_fused bit 5: the block LayerNorm rides in the prologue of the fused GRN-MLP passes (csrc/mlp.hip) — no LayerNorm
        # pass over y, and in inference no normalised rows in memory at all
        ln_in = fused and bool(self._mlp_flag() & 32) and hasattr(o, "mlp_fc1_ln")
        if ln_in:
            xh, rstd = y, None
        else:
            xh, _, rstd = o.ln_fwd(y, None, None, M, C, 1e-6, need_mean=False)
            del y
        colsq = self._za.take(B, 4 * C)
        # stochastic depth (timm DropPath, scale_by_keep): the whole branch of a sample is dropped with probability dp and
        # the survivors are scaled by 1 / (1 - dp); training mode only.  One device-side draw per block (graph-capturable).
        dpm = None
        if w.dp > 0.0 and self.model.training:
            inj = self._dp_inject
            dpm = inj.pop(0) if inj else (torch.rand(B, device=x.device) < (1.0 - w.dp)).float() / (1.0 - w.dp)
        lne = 1e-6 if ln_in else 0.0
        if fused and save is None:
            # inference: the 4C-wide hidden never leaves the CU (csrc/mlp.hip) — pass 1 = GRN statistics, pass 2 = fc1
            # recomputed, GELU, GRN, fc2, bias, shortcut
            if w.img is None:
                w.img = o.mlp_pack(w.W1f, w.W2, C)
                o.flush()  # inside an open task-list batch the pack is only queued; its reader is the next launch
            if ln_in:
                o.mlp_stats(xh, w.img, w.b1f, colsq, M, C, hw, ln_eps=lne)
                s = o.grn_scale(colsq, w.grn_w)
                out = o.mlp_out(xh, w.img, w.b1f, s, w.grn_b, w.fc2_b, xres, dpm, M, C, hw, ln_eps=lne)
            else:
                o.mlp_stats(xh, w.img, w.b1f, colsq, M, C, hw)
                s = o.grn_scale(colsq, w.grn_w)
                out = o.mlp_out(xh, w.img, w.b1f, s, w.grn_b, w.fc2_b, xres, dpm, M, C, hw)
            if rows is not None:
                out = o.rows_select(out, rows[1], B * H * Wd, C)
            return out
        # fc1 writes the pre-activation h (needed for gelu' in backward) AND the activation g = gelu(h):
        # fc2, the fc2 weight gradient and the GRN statistics path all consume g, so GELU is evaluated once
        # (inference keeps the activation only: C = NULL skips the pre-activation store, a third of the block's 4C-wide traffic)
        chunk = 0
        if save is not None and rows is None and ln_in and self._mlp_mode(C, hw, M, dt, True) and self._mlp_drop_xh(C, hw, M, dt, B) \
                and C > 64 and hw % 128 == 0 and hw // 128 >= 8:
            chunk = self._sample_chunk(B, hw, C, "fwd")
        if chunk:
            # round 6: sample-chunk-major forward of the block's two GEMM-shaped launches — fc1 (MODE 6) writes the activation g
            # of `chunk` samples, the per-sample GRN scale of exactly those samples follows, and fc2 reads that g chunk while it
            # is still in the Infinity Cache (g is needed again in the backward, so the whole tensor is kept as before)
            if w.img is None:
                w.img = o.mlp_pack(w.W1f, w.W2, C)
                o.flush()
            mean = torch.empty(M, dtype=torch.float32, device=x.device)
            rstd = torch.empty(M, dtype=torch.float32, device=x.device)
            gact = torch.empty((M, 4 * C), dtype=dt, device=x.device)
            s = torch.empty_like(colsq)
            out = torch.empty((M, C), dtype=dt, device=x.device)
            b2 = w.b2f
            if b2 is None:
                b2 = o.matvec(w.fc2_w, w.grn_b, w.fc2_b, C, 4 * C)
                o.flush()
            for b0 in range(0, B, chunk):
                b1_ = min(B, b0 + chunk)
                r0, r1 = b0 * hw, b1_ * hw
                o.mlp_fc1_ln(y[r0:r1], w.img, w.b1f, colsq[b0:b1_], r1 - r0, C, hw, 1e-6, store_h=False, store_xh=False,
                             outs=(mean[r0:r1], rstd[r0:r1], gact[r0:r1]))
                o.grn_scale(colsq[b0:b1_], w.grn_w, out=s[b0:b1_])
                Ws = o.scale_weight_samples(w.fc2_w, s[b0:b1_], dt)
                o.gemm("nt", gact[r0:r1], Ws, out[r0:r1], r1 - r0, C, 4 * C, 4 * C, 4 * C, C, dtype=dt, hw=hw, b_bstride=C * 4 * C,
                       epi=L.EPI_BIAS_RES, bias=b2, res=xres[r0:r1], ldr=C, rscale=None if dpm is None else dpm[b0:b1_])
            save.append((x, (y, mean), rstd, None, gact, colsq, s, rows, dpm))
            return out
        if save is not None and self._mlp_mode(C, hw, M, dt, True):
            # training fc1 on the fused kernel's statistics pass, which also stores h and g (csrc/mlp.hip MODE 2)
            if w.img is None:
                w.img = o.mlp_pack(w.W1f, w.W2, C)
                o.flush()
            # mlp_fused bit 6: the pre-activation h is NOT stored — the backward's dh pass recomputes it from the C-wide
            # normalised rows (csrc/mlp.hip MODE 6 here, MODE 5 there): one 4C-wide write and one 4C-wide read less per block
            keep_h = not self._mlp_recompute_h(C, hw, M, dt, B)
            if ln_in:
                # bit 7: ... and neither is x^ — `xh` is then the pair (y, row means) the backward re-normalises from
                xh, rstd, h, gact = o.mlp_fc1_ln(xh, w.img, w.b1f, colsq, M, C, hw, 1e-6, store_h=keep_h,
                                                 store_xh=keep_h or not self._mlp_drop_xh(C, hw, M, dt, B))
            else:
                h, gact = o.mlp_fc1(xh, w.img, w.b1f, colsq, M, C, hw, store_h=keep_h)
        else:
            h = torch.empty((M, 4 * C), dtype=dt, device=x.device) if save is not None else None
            gact = torch.empty((M, 4 * C), dtype=dt, device=x.device)
            o.gemm("nt", xh, w.W1f, h, M, 4 * C, C, C, C, 4 * C, dtype=dt, epi=L.EPI_BIAS_GELU_SQ, bias=w.b1f, red0=colsq,
                   hw=hw, C2=gact)
        s = o.grn_scale(colsq, w.grn_w)
        out = torch.empty((M, C), dtype=dt, device=x.device)
        if dt == torch.bfloat16 and C > 64 and hw % 128 == 0 and hw // 128 >= 8:
            # large feature maps: fold the GRN affine into per-sample fc2 weights,
            #   (g·s_b + β)·W2ᵀ = g·(W2·diag(s_b))ᵀ + W2·β,
            # so fc2 is a plain GEMM (the operand prologue costs +60 % on these launches); B·C·4C extra weight bytes
            # are small next to the M·4C activation bytes when a sample spans >= 8 row tiles
            Ws = o.scale_weight_samples(w.fc2_w, s, dt)
            b2 = w.b2f
            if b2 is None:
                b2 = o.matvec(w.fc2_w, w.grn_b, w.fc2_b, C, 4 * C)
                o.flush()
            o.gemm("nt", gact, Ws, out, M, C, 4 * C, 4 * C, 4 * C, C, dtype=dt, hw=hw, b_bstride=C * 4 * C,
                   epi=L.EPI_BIAS_RES, bias=b2, res=xres, ldr=C, rscale=dpm)
        else:
            o.gemm("nt", gact, w.W2, out, M, C, 4 * C, 4 * C, 4 * C, C, dtype=dt, pro=L.PRO_GRN, grn_s=s,
                   grn_b=w.grn_b, hw=hw, epi=L.EPI_BIAS_RES, bias=w.fc2_b, res=xres, ldr=C, rscale=dpm)
        if rows is not None:
            out = o.rows_select(out, rows[1], B * H * Wd, C)  # masked_unpatchify: zero rows where masked (shortcut is 0 there)
        if save is not None:
            save.append((x, xh, rstd, h, gact, colsq, s, rows, dpm))
        return out

    def _block_bwd(self, dout, w, saved, B, H, Wd, dt):
        o, g = self.ops, self.g
        C, M = w.C, B * H * Wd
        blk = w.p
        x, xh, rstd, h, gact, colsq, s, rows, dpm = saved
        dev = dout.device
        hw = H * Wd
        dfull = dout
        if rows is not None:  # masked path: the MLP branch only sees the gradient of the kept tokens
            idx, inv, keep, Lr = rows
            M, hw = B * Lr, Lr
            dout = o.rows_select(dfull, idx, M, C)
        if dpm is not None:  # the branch saw dout * (0 | 1/keep) of its sample; the shortcut keeps the full gradient (dfull)
            dout = o.scale_rows_samples(dout, dpm, M, C, hw)
        if w.v1:  # gradients of the folded (Ws, bs) and of the (constant, zero) GRN land in scratch
            dW2, db2 = self._za.take(C, 4 * C), self._za.take(C)
            dgw, dgb = self._za.take(4 * C), self._za.take(4 * C)
        else:
            dW2, db2 = g(blk.mlp.fc2.weight), g(blk.mlp.fc2.bias)
            dgw, dgb = g(blk.mlp.grn.weight), g(blk.mlp.grn.bias)
        fused_bwd = self._mlp_bwd_fused(C, hw, M, dt, B)
        if h is None and not fused_bwd:
            raise RuntimeError("this block's forward did not store the pre-activation h (mlp_fused bit 6) and the fused backward "
                               "that recomputes it is switched off now: do not change `mlp_fused` between a forward and its backward")
        PS = self._za.take(2, B, 4 * C)
        if fused_bwd == 1:
            # dz = dout·W2 is LINEAR in dout, so everything the backward needs from "Σ over the sample of dz·(something)" comes
            # out of the per-sample products Q_b = dout_bᵀ·g_b that the weight gradient computes anyway:
            #   P_b = Σ_hw dz·g = Σ_c W2[c,:]·Q_b[c,:],  S_b = Σ_hw dz = Σ_c W2[c,:]·cs_b[c],  dW2 = Σ_b s_b·Q_b + cs⊗β
            # — one TN GEMM without operand prologue (30 % faster than the prologue form) and a tiny reduction replace the
            # fc2 weight-gradient GEMM AND the statistics pass over the 4C-wide tensors; dz is never formed for them
            Qb = torch.empty((B, C, 4 * C), dtype=torch.float32, device=dev)
            csb = torch.empty((B, C), dtype=torch.float32, device=dev)
            o.gemm("tn", gact, dout, Qb, M, C, 4 * C, 4 * C, C, 4 * C, dtype=dt, hw=hw, colsum=csb, b_bstride=C * 4 * C)
            o.grn_q_reduce(Qb, csb, w.W2, s, w.grn_b, PS[0], PS[1], dW2, db2)
            del Qb
        stats_in_tn = False
        if fused_bwd == 1:
            pass
        elif fused_bwd == 2 and hasattr(o, "tn_grn_stats_ok") and o.tn_grn_stats_ok(M, C, 4 * C, hw, dt):
            # round 5: the products are too large to store (C = 384 at B = 512), but each per-sample TILE of Q_b passes through the
            # accumulators of the fc2 weight-gradient GEMM anyway — that launch scales it by s_b into dW2 and contracts it with its
            # W2 tile into P_b on the way (csrc/gemm.hip, gemm_tn_fast_kernel PRO == 2).  The statistics pass that recomputed dz for
            # P and S (csrc/mlp.hip MODE 3, a full M x 4C x C contraction per block) is gone; S only ever fed dbeta = sum_b S_b =
            # (column sums of dout) . W2, a matvec on the sums this launch produces for the bias gradient.
            # ConvNeXt-V1 blocks: db2 is zeroed scratch that `layer_scale_unfold` reads in a DIRECT launch right below, while
            # `transpose_f32` is only queued inside the backward segment's task-list batch — the GEMM writes the column sums into
            # db2 itself there (fc2.bias gradient and its share of dgamma were lost on v1 backbones at this path)
            cs = db2 if w.v1 else self._za.take(C)
            o.gemm("tn", gact, dout, dW2, M, C, 4 * C, 4 * C, C, 4 * C, dtype=dt, pro=L.PRO_GRN, grn_s=s, grn_b=w.grn_b, hw=hw,
                   colsum=cs, aux=w.W2, ldx=4 * C, red0=PS[0])
            if not w.v1:
                o.transpose_f32(cs, db2, C, 1, True)     # db2 += cs
            o.matvec_t_add(w.fc2_w, cs, dgb, C, 4 * C)   # dbeta += W2^T cs
            stats_in_tn = True
        else:
            # fc2: weight gradient (Z recomputed in the operand prologue) + bias gradient
            o.gemm("tn", gact, dout, dW2, M, C, 4 * C, 4 * C, C, 4 * C, dtype=dt, pro=L.PRO_GRN, grn_s=s,
                   grn_b=w.grn_b, hw=hw, colsum=db2)
        if w.v1:
            o.layer_scale_unfold(dW2, db2, blk.mlp.fc2.weight, blk.mlp.fc2.bias, blk.gamma, g(blk.mlp.fc2.weight),
                                 g(blk.mlp.fc2.bias), g(blk.gamma))
        ln_re = isinstance(xh, tuple)  # (y, mean): the forward stored no normalised rows (mlp_fused bit 7)
        if ln_re and not (fused_bwd and h is None and self._mlp_drop_xh(C, hw, M, dt, B)):
            raise RuntimeError("this block's forward stored no normalised rows (mlp_fused bit 7) and the passes that re-normalise "
                               "are switched off now: do not change `mlp_fused` between a forward and its backward")
        u1 = None
        if ln_re:
            cs2 = self._za.take(2, 4 * C)  # {column sums of dh, u = sum_r dh' * mean}
            db1f, u1 = cs2[0], cs2[1]
        else:
            db1f = self._za.take(4 * C)
        if fused_bwd:
            # the 4C-wide dz is never written: the statistics came from the per-sample products above; this pass recomputes
            # dz = dout·W2 tile by tile (K = C is short) and writes dh directly (csrc/mlp.hip MODE 4) — one 4C-wide write
            # where the unfused pair (dz GEMM, then GRN / GELU backward over it) has two
            img2 = w.img2
            if img2 is None:
                # not packed by prepare() (mlp_fused bit 3 switched on between a forward and its backward): _block_bwd runs
                # inside the backward segment's open task-list batch, where the pack would only be QUEUED while the passes below
                # read the image right away — launch the list before going on
                img2 = w.img2 = o.mlp_pack(w.W2T, w.W2, C)
                o.flush()
            if fused_bwd == 2 and not stats_in_tn:  # statistics by recomputing dz tile by tile (P = Σ dz·g, S = Σ dz), nothing stored
                o.mlp_bwd_stats(dout, img2, gact, PS[0], PS[1], M, C, hw)
            if stats_in_tn:
                t = o.grn_bwd_stats(colsq, PS[0], w.grn_w, dgw)
            else:
                t = o.grn_bwd_stats(colsq, PS[0], w.grn_w, dgw, Sb=PS[1], dbeta=dgb)
            chunk = self._sample_chunk(B, hw, C) if ln_re and rows is None else 0
            if chunk:
                # round 6: sample-chunk-major schedule of the three launches that touch the 4C-wide dh — the
                # dh pass writes a chunk of `chunk` samples into ONE reused buffer sized for the 256 MiB Infinity Cache, and its two
                # readers (fc1 data gradient with the LayerNorm backward, fc1 weight gradient) run right behind it, so that both
                # re-reads are served memory-side instead of from HBM; the full [M, 4C] dh is never allocated
                buf = torch.empty((chunk * hw, 4 * C), dtype=dt, device=dev)
                dy = torch.empty((M, C), dtype=dt, device=dev)
                dW1f = self._za.take(4 * C, C)
                y_, mean_ = xh
                for b0 in range(0, B, chunk):
                    b1_ = min(B, b0 + chunk)
                    r0, r1 = b0 * hw, b1_ * hw
                    dzc = o.mlp_bwd_dh_ln(dout[r0:r1], y_[r0:r1], mean_[r0:r1], rstd[r0:r1], img2, w.img, w.b1f, s[b0:b1_], t[b0:b1_],
                                          cs2, r1 - r0, C, hw, out=buf[: r1 - r0])
                    o.dgrad_ln_bwd(dzc, w.W1fT, y_[r0:r1], rstd[r0:r1], r1 - r0, C, 4 * C, mean=mean_[r0:r1], out=dy[r0:r1])
                    o.gemm("tn", y_[r0:r1], dzc, dW1f, r1 - r0, 4 * C, C, C, 4 * C, C, dtype=dt)
                del buf, dzc
                dz = None
            elif ln_re:  # h recomputed from y re-normalised on chip; dz holds dh * rstd (MODE 7)
                dz = o.mlp_bwd_dh_ln(dout, xh[0], xh[1], rstd, img2, w.img, w.b1f, s, t, cs2, M, C, hw)
            elif h is None:  # h recomputed on chip from the normalised rows (MODE 5)
                dz = o.mlp_bwd_dh_re(dout, xh, img2, w.img, w.b1f, s, t, db1f, M, C, hw)
            else:
                dz = o.mlp_bwd_dh(dout, img2, h, s, t, db1f, M, C, hw)  # (named dz below: it holds dH)
        else:
            # fc2 data gradient dZ, with Σ dZ·gelu(h) (GRN statistics path) and Σ dZ (GRN beta gradient)
            dz = torch.empty((M, 4 * C), dtype=dt, device=dev)
            o.gemm("nt", dout, w.W2T, dz, M, 4 * C, C, C, C, 4 * C, dtype=dt, epi=L.EPI_DZ, aux=gact, ldx=4 * C, red0=PS[0],
                   red1=PS[1], hw=hw)
            t = o.grn_bwd_stats(colsq, PS[0], w.grn_w, dgw, Sb=PS[1], dbeta=dgb)
            o.grn_gelu_bwd(dz, h, s, t, db1f, M, 4 * C, hw)  # dz now holds dH
        # fc1 data gradient; where one column tile spans the row (C <= 256) the block LayerNorm's backward rides in the GEMM's
        # epilogue (VSX_EPI_LN_BWD): dx^ is never written, the LayerNorm-backward launch and two C-wide passes go
        chunked = fused_bwd and ln_re and dz is None
        if chunked:
            xh = xh[0]
        elif ln_re:
            # dz = dh * rstd: the accumulator is rstd * dx^ already, x^ is re-formed from y in the epilogue; the weight gradient
            # contracts dz with y itself, dh^T x^ = dz^T y - u (x) 1 — the rank-1 term goes in the unfold below
            dy = o.dgrad_ln_bwd(dz, w.W1fT, xh[0], rstd, M, C, 4 * C, mean=xh[1])
            xh = xh[0]
        else:
            dy = o.dgrad_ln_bwd(dz, w.W1fT, xh, rstd, M, C, 4 * C) if hasattr(o, "dgrad_ln_bwd") else None
        dxh = None
        if dy is None:
            dxh = torch.empty((M, C), dtype=dt, device=dev)
            o.gemm("nt", dz, w.W1fT, dxh, M, C, 4 * C, 4 * C, 4 * C, C, dtype=dt)
        if not chunked:
            dW1f = self._za.take(4 * C, C)
            o.gemm("tn", xh, dz, dW1f, M, 4 * C, C, C, 4 * C, C, dtype=dt)
        del dz
        # unfold the LayerNorm affine: dW1 = dW1f·diag(γ) + db1f ⊗ β, dγ = Σ_r dW1f ⊙ W1, dβ = W1ᵀ db1f, db1 = db1f
        o.unprep_grad(dW1f, g(blk.mlp.fc1.weight), 4 * C, C, 1, gamma=blk.norm.weight, W=blk.mlp.fc1.weight,
                      dgamma=g(blk.norm.weight), u=db1f, beta=blk.norm.bias, rowsub=u1)
        o.matvec_t_add(blk.mlp.fc1.weight, db1f, g(blk.norm.bias), 4 * C, C)
        o.transpose_f32(db1f, g(blk.mlp.fc1.bias), 4 * C, 1, True)  # g(b1) += db1f (a [4C, 1] "transpose": no ATen launch in the step)
        if dy is None:
            dy = o.ln_bwd(dxh, xh, None, rstd, None, None, None, None, M, C)
        del dxh
        if rows is not None:
            dy = o.rows_select(dy, inv, B * H * Wd, C)  # adjoint of the gather: zero-filled scatter
        dx = o.dwconv7_bwd_data(dy, w.dw_w, dfull, B, H, Wd, C)
        ddw = self._za.take(49, C)
        o.dwconv7_bwd_weight(dy, x, ddw, g(blk.conv_dw.bias), B, H, Wd, C)
        o.transpose_f32(ddw, g(blk.conv_dw.weight), 49, C, True)
        if rows is not None:
            dx = o.rows_select(dx, keep, B * H * Wd, C)  # adjoint of `x *= unmasked` (shortcut and dwconv input alike)
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, dt: torch.dtype, need_bwd: bool, masks=None, bn_groups: int = 1):
        """A forward WITHOUT a backward behind it (predict / validation / `InferStep` captures) forms the per-sample GRN and
        InstanceNorm sums in a fixed order (`det_reduce`), as the reference's CPU path is deterministic: with fp32 atomics the
        bf16 forward at 2048^2 differs by 7e-3 of the output maximum from run to run, the fixed order costs 0.3 % of a pass
        (profiles/r05_det_reduce.txt).  Training forwards keep the atomics unless the flag is set by hand;
        ``VSX_DET_INFER=0`` in the environment switches the automatic setting off."""
        auto = (not need_bwd) and self.ops.__name__.endswith("viscy_amd.ops") and os.environ.get("VSX_DET_INFER", "1") != "0" \
            and L.lib().vsx_get_flag(b"det_reduce") == 0
        if not auto:
            return self._forward(x, dt, need_bwd, masks, bn_groups)
        L.lib().vsx_set_flag(b"det_reduce", 1)
        try:
            return self._forward(x, dt, need_bwd, masks, bn_groups)
        finally:
            L.lib().vsx_set_flag(b"det_reduce", 0)

    def _forward(self, x: Tensor, dt: torch.dtype, need_bwd: bool, masks=None, bn_groups: int = 1):
        """``masks``: None (dense) or, per encoder stage, ``(idx, inv, keep, L)`` int32 row maps of the FCMAE mask at that
        stage's resolution (see ``viscy_amd.fcmae.stage_row_maps``)."""
        o, cfg, m = self.ops, self.cfg, self.model
        W = self.prepare(dt, need_bwd)
        B, Cin, Z, H, Wd = x.shape
        za_key = ("fwd", B, H, Wd, masks is not None)
        self._za = za = _ZeroArena(x.device, self._za_need.get(za_key, 0), self.ops)
        # FCMAE 2-D branch (fcmae.py:369-370): the reference runs conv2d whenever x.shape[2] == 1, whatever in_stack_depth is —
        # a 2-D FCMAE (in_stack_depth = 1) trains and loads its conv2d weights, never the conv3d ones
        flat_stem = Z == 1 and "stem2d_W" in W
        if Cin != cfg["in_channels"] or (Z != cfg["in_stack_depth"] and not flat_stem):
            raise ValueError(f"expected input (B,{cfg['in_channels']},{cfg['in_stack_depth']},Y,X), got {tuple(x.shape)}")
        kz, ky, kx = cfg["stem_kernel"]
        if H % (8 * ky) or Wd % (8 * kx):
            raise ValueError(f"Y and X must be divisible by {8 * ky} (got {H}x{Wd}); VSUNet pads to a multiple of 64")
        h, w = H // ky, Wd // kx
        dims = cfg["dims"]
        C0 = dims[0]
        # the prepared operands travel with the forward that used them: a later (no_grad / other dtype) forward re-prepares
        # self.W, and the backward of THIS forward must still see its own transposed weights
        sv = {"shape": (B, H, Wd), "dt": dt, "masked": masks is not None, "W": W} if need_bwd else None
        if need_bwd:
            self._pending_bwd += 1
        # ---- stem: patch gather + projection GEMM, then encoder stem_1 LayerNorm2d
        stem_W, stem_b = (W["stem2d_W"], W["stem2d_b"]) if flat_stem else (W["stem_W"], W["stem_b"])
        P = o.stem_im2col(x.contiguous(), (1 if flat_stem else kz, ky, kx), dt, ld=stem_W.shape[1])
        M0, K0 = B * h * w, P.shape[1]
        f = torch.empty((M0, C0), dtype=dt, device=x.device)
        o.gemm("nt", P, stem_W, f, M0, C0, K0, K0, K0, C0, dtype=dt, epi=L.EPI_BIAS, bias=stem_b)
        ln1 = m.encoder_stages.stem_1
        cur, mean, rstd = o.ln_fwd(f, ln1.weight, ln1.bias, M0, C0)
        if need_bwd:
            sv["stem"] = (P, f, mean, rstd)
            sv["flat_stem"] = flat_stem
        else:
            del P, f
        # ---- encoder
        feats, enc_sv = [], []
        ch, cw, cc = h, w, C0
        for i, (proj, blocks) in enumerate(W["enc"]):
            st_sv = {"blocks": []}
            if proj is not None:
                M_in = B * ch * cw
                xn, mean, rstd = o.ln_fwd(cur, proj.ln.weight, proj.ln.bias, M_in, cc)
                ch, cw = ch // 2, cw // 2
                nxt = torch.empty((B * ch * cw, proj.cout), dtype=dt, device=x.device)
                o.gemm("nt", xn, proj.W, nxt, B * ch * cw, proj.cout, 4 * cc, cc, 4 * cc, proj.cout, dtype=dt,
                       a_mode=L.A_PATCH2, gh=ch, gw=cw, cs=cc, epi=L.EPI_BIAS, bias=proj.conv.bias)
                st_sv["proj"] = (cur, xn, mean, rstd)
                cur, cc = nxt, proj.cout
            rows = None
            if masks is not None:
                rows = masks[i]
                # first block of the stage: `x *= unmasked` (fcmae.py:216-217); later blocks receive masked maps already
                cur = o.rows_select(cur, rows[2], B * ch * cw, cc)
            for bw in blocks:
                cur = self._block_fwd(cur, bw, B, ch, cw, dt, st_sv["blocks"] if need_bwd else None, rows)
            feats.append((cur, ch, cw, cc))
            enc_sv.append(st_sv)
        if cfg.get("head") == "embed":
            out = self._embed_tail_fwd(feats[3], B, sv, bn_groups)
            if need_bwd:
                sv["enc"] = enc_sv
                sv["feat_dims"] = [(a, b_, c) for (_, a, b_, c) in feats]
            self._za_need[za_key] = za.used
            return out, sv
        # ---- decoder
        dec_sv = []
        feat, fh, fw, fc = feats[3]
        for k, (proj, blocks) in enumerate(W["dec"]):
            skip, sh, sw_, sc = feats[2 - k]
            assert sh == 2 * fh and sw_ == 2 * fw
            c_up = fc // 4
            pre = W["dec_pre"][k]
            low_in = None
            if pre is not None:  # MONAI SubpixelUpsample conv_block: dense 3x3 convolution at the low resolution
                pc, Wp, _ = pre
                col = o.im2col3x3(feat, B, fh, fw, fc)
                low_in = feat
                feat = torch.empty((B * fh * fw, fc), dtype=dt, device=x.device)
                o.gemm("nt", col, Wp, feat, B * fh * fw, fc, 9 * fc, 9 * fc, 9 * fc, fc, dtype=dt, epi=L.EPI_BIAS, bias=pc.bias)
                del col
            cat = o.pixel_shuffle_cat_fwd(feat, skip, B, fh, fw, c_up, sc)
            fh, fw = sh, sw_
            Mk, ccat = B * fh * fw, c_up + sc
            xn, mean, rstd = o.ln_fwd(cat, proj.ln.weight, proj.ln.bias, Mk, ccat)
            cur = torch.empty((Mk, proj.cout), dtype=dt, device=x.device)
            o.gemm("nt", xn, proj.W, cur, Mk, proj.cout, ccat, ccat, ccat, proj.cout, dtype=dt, epi=L.EPI_BIAS,
                   bias=proj.conv.bias)
            st_sv = {"proj": (cat, xn, mean, rstd, c_up, sc), "blocks": [], "pre_in": low_in}
            for bw in blocks:
                cur = self._block_fwd(cur, bw, B, fh, fw, dt, st_sv["blocks"] if need_bwd else None)
            feat, fc = cur, proj.cout
            dec_sv.append(st_sv)
        # ---- head
        if cfg.get("head", "conv") == "shuffle":
            # PixelToVoxelShuffleHead (heads.py:656-685): pixel shuffle x s + pad-pool + reshape, parameter free
            sxy = cfg["stem_kernel"][-1]
            out = o.voxel_shuffle_fwd(feat, B, fh, fw, cfg["out_channels"], cfg["out_stack_depth"], sxy, True)
            if need_bwd:
                sv["enc"], sv["dec"] = enc_sv, dec_sv
                sv["feat_dims"] = [(a, b_, c) for (_, a, b_, c) in feats]
                sv["head"] = (fh, fw)
            self._za_need[za_key] = za.used
            return out, sv
        Zo, D7 = cfg["out_stack_depth"], cfg["out_stack_depth"] + 2
        hc = m.head.conv[0].conv
        cmid, c3 = hc.weight.shape[0], hc.weight.shape[1]
        cout = cfg["out_channels"]
        hdt = self._head_dtype(dt)
        hin = o.head_shuffle_fwd(feat if hdt == dt else feat.float(), B, fh, fw, c3, D7, cfg["head_pool"])
        H2, W2 = 2 * fh, 2 * fw
        Mh = B * H2 * W2
        stats = self._za.take(2, B, cmid)
        direct = o.head_conv_supported(H2, W2, c3, cmid, Zo, hdt)
        if direct:
            U = o.head_conv_fwd(hin, W["head_Wc"], hc.bias, stats[0], stats[1], B, H2, W2, c3, cmid, Zo)
        else:
            U = torch.empty((Mh, Zo * cmid), dtype=hdt, device=x.device)
            o.gemm_z("nt", hin, W["head_Wc"], U, Mh, cmid, 27 * c3, D7 * c3, 27 * c3, Zo * cmid, dtype=hdt,
                     a_mode=L.A_CONV3, gh=H2, gw=W2, cs=3 * c3, nz=Zo, a_coff=[z * c3 for z in range(Zo)],
                     b_off=[0] * Zo, c_coff=[z * cmid for z in range(Zo)], epi=L.EPI_BIAS_STATS, bias=hc.bias,
                     red0=stats[0], red1=stats[1], hw=H2 * W2)
        w2 = m.head.conv[1].weight.view(4 * cout, cmid)
        out = o.head_out_fwd(U, stats[0], stats[1], w2, m.head.conv[1].bias, m.head.conv[0].adn.A.weight, B, H2, W2, Zo,
                             cmid, cout)
        if need_bwd:
            sv["enc"], sv["dec"] = enc_sv, dec_sv
            sv["feat_dims"] = [(a, b_, c) for (_, a, b_, c) in feats]
            sv["head"] = (hin, U, stats, fh, fw)
        self._za_need[za_key] = za.used
        return out, sv

    # ------------------------------------------------------------------ ContrastiveEncoder tail (contrastive/encoder.py:93-154)
    def _bn_groups_fwd(self, z, bn, training, relu, groups):
        """BatchNorm1d over ``groups`` consecutive row blocks, each with its own batch statistics and its own running-stat
        update, in order (== that many separate forward calls of the module)"""
        o = self.ops
        n = z.shape[0] // groups
        ys, sms, srs = [], [], []
        for gi in range(groups):
            y, sm, sr = o.bn1d_fwd(z[gi * n:(gi + 1) * n], bn.weight, bn.bias, bn.running_mean, bn.running_var, training, relu)
            ys.append(y); sms.append(sm); srs.append(sr)
        return (ys[0] if groups == 1 else torch.cat(ys)), sms, srs

    def _bn_groups_bwd(self, dy, z, y, bn, sms, srs, training, relu):
        o, g = self.ops, self.g
        groups = len(sms)
        n = z.shape[0] // groups
        dx = [o.bn1d_bwd(dy[gi * n:(gi + 1) * n], z[gi * n:(gi + 1) * n], y[gi * n:(gi + 1) * n], bn.weight, sms[gi], srs[gi],
                         g(bn.weight), g(bn.bias), training, relu) for gi in range(groups)]
        return dx[0] if groups == 1 else torch.cat(dx)

    def _embed_tail_fwd(self, feat3, B, sv, bn_groups: int = 1):
        """global average pool -> LayerNorm (timm head.norm) = embedding; Linear -> BN -> ReLU -> Linear -> BN = projection.
        fp32 throughout ([B, 768]-sized tensors); BatchNorm statistics are those of THIS call's batch (the reference runs
        anchor and positive through separate forwards, dynaclr/engine.py:265-266); ``bn_groups = 2`` gives a concatenated
        [anchor; positive] batch exactly those per-call statistics while the trunk runs once."""
        o, m, W = self.ops, self.model, self.W
        feat, fh, fw, fc = feat3
        t = m.tail
        training = bool(m.training)
        pooled = o.avgpool_rows_fwd(feat, B, fh * fw, fc)
        emb, mean, rstd = o.ln_fwd(pooled, t.norm.weight, t.norm.bias, B, fc)
        E, P = t.fc0.weight.shape[0], t.fc3.weight.shape[0]
        z0 = torch.empty((B, E), dtype=torch.float32, device=feat.device)
        o.gemm("nt", emb, W["fc0"], z0, B, E, fc, fc, fc, E, dtype=torch.float32, epi=L.EPI_BIAS, bias=t.fc0.bias)
        if B % bn_groups:
            raise ValueError(f"batch {B} is not divisible into {bn_groups} BatchNorm groups")
        y1, sm1, sr1 = self._bn_groups_fwd(z0, t.bn1, training, True, bn_groups)
        z3 = torch.empty((B, P), dtype=torch.float32, device=feat.device)
        o.gemm("nt", y1, W["fc3"], z3, B, P, E, E, E, P, dtype=torch.float32, epi=L.EPI_BIAS, bias=t.fc3.bias)
        y4, sm4, sr4 = self._bn_groups_fwd(z3, t.bn4, training, False, bn_groups)
        if training:
            t.bn1.num_batches_tracked += bn_groups
            t.bn4.num_batches_tracked += bn_groups
        if sv is not None:
            sv["tail"] = (pooled, mean, rstd, emb, z0, y1, sm1, sr1, z3, y4, sm4, sr4, training, fh, fw, fc)
        return emb, y4

    def _embed_tail_bwd(self, sv, demb, dproj, dt, B):
        o, m, W, g = self.ops, self.model, sv["W"], self.g
        pooled, mean, rstd, emb, z0, y1, sm1, sr1, z3, y4, sm4, sr4, training, fh, fw, fc = sv["tail"]
        t = m.tail
        E, P = t.fc0.weight.shape[0], t.fc3.weight.shape[0]
        dev = emb.device
        d_emb = demb.contiguous().float() if demb is not None else None
        if dproj is not None:
            dz3 = self._bn_groups_bwd(dproj.contiguous().float(), z3, y4, t.bn4, sm4, sr4, training, False)
            o.gemm("tn", y1, dz3, g(t.fc3.weight), B, P, E, E, P, E, dtype=torch.float32, colsum=g(t.fc3.bias))
            dy1 = torch.empty((B, E), dtype=torch.float32, device=dev)
            o.gemm("nt", dz3, W["fc3T"], dy1, B, E, P, P, P, E, dtype=torch.float32)
            dz0 = self._bn_groups_bwd(dy1, z0, y1, t.bn1, sm1, sr1, training, True)
            o.gemm("tn", emb, dz0, g(t.fc0.weight), B, E, fc, fc, E, fc, dtype=torch.float32, colsum=g(t.fc0.bias))
            de = torch.empty((B, fc), dtype=torch.float32, device=dev)
            o.gemm("nt", dz0, W["fc0T"], de, B, fc, E, E, E, fc, dtype=torch.float32)
            d_emb = de if d_emb is None else d_emb + de
        if d_emb is None:
            d_emb = torch.zeros((B, fc), dtype=torch.float32, device=dev)
        dpooled = o.ln_bwd(d_emb, pooled, mean, rstd, t.norm.weight, None, g(t.norm.weight), g(t.norm.bias), B, fc)
        return o.avgpool_rows_bwd(dpooled, B, fh * fw, fc, dt)

    def _head_conv_bwd(self, sv, dout, dt, B, dev):
        """PixelToVoxelHead backward; returns the gradient of the decoder feature map."""
        o, cfg, m, g, W = self.ops, self.cfg, self.model, self.g, sv["W"]
        Zo, D7 = cfg["out_stack_depth"], cfg["out_stack_depth"] + 2
        hc = m.head.conv[0].conv
        cmid, c3 = hc.weight.shape[0], hc.weight.shape[1]
        cout = cfg["out_channels"]
        hin, U, stats, fh, fw = sv["head"]
        H2, W2 = 2 * fh, 2 * fw
        Mh = B * H2 * W2
        alpha = m.head.conv[0].adn.A.weight
        w2 = m.head.conv[1].weight.view(4 * cout, cmid)
        # ---- head
        S = self._za.take(2, B, cmid)
        if dt == torch.bfloat16 and (cmid, cout) in ((32, 2), (64, 4)):
            # pass 1 with the 1x1x1 weight gradient folded in (voxel contraction on MFMA inside the kernel): the
            # [M5, cmid] activation is never written and the skinny TN GEMM over it disappears
            dv = o.head_out_bwd1_wgrad(U, stats[0], stats[1], w2, alpha, dout.contiguous().float(), S[0], S[1], g(alpha),
                                       g(m.head.conv[1].weight), g(m.head.conv[1].bias), B, H2, W2, Zo, cmid, cout)
        else:
            act, dv = o.head_out_bwd1(U, stats[0], stats[1], w2, alpha, dout.contiguous().float(), S[0], S[1], g(alpha), B,
                                      H2, W2, Zo, cmid, cout)
            o.gemm("tn", act, dv, g(m.head.conv[1].weight), Mh * Zo, 4 * cout, cmid, cmid, 4 * cout, cmid, dtype=dt,
                   colsum=g(m.head.conv[1].bias))
            del act
        dU = o.head_out_bwd2(U, stats[0], stats[1], w2, alpha, dv, S[0], S[1], B, H2, W2, Zo, cmid, cout)
        del dv
        dWc = self._za.take(cmid, 27 * c3)
        direct = "head_Wp" in W and o.head_conv_supported(H2, W2, c3, cmid, Zo, dt)
        if direct:
            o.head_conv_wgrad(hin, dU, dWc, g(hc.bias), B, H2, W2, c3, cmid, Zo)
        else:
            o.gemm_z("tn", hin, dU, dWc, Mh, cmid, 27 * c3, D7 * c3, Zo * cmid, 27 * c3, dtype=dt, a_mode=L.A_CONV3, gh=H2,
                     gw=W2, cs=3 * c3, nz=Zo, a_coff=[z * c3 for z in range(Zo)], b_off=[z * cmid for z in range(Zo)],
                     c_coff=[0] * Zo, colsum=g(hc.bias))
        o.unprep_grad(dWc, g(hc.weight), cmid, c3, 27, tapmode=1)
        if direct:
            dhin = o.head_conv_dgrad(dU, W["head_Wp"], B, H2, W2, c3, cmid, Zo)
        else:
            dhin = torch.empty((Mh, D7 * c3), dtype=dt, device=dev)
            zs = [min(max(zp - 2, 0), Zo - 3) for zp in range(D7)]
            o.gemm_z("nt", dU, W["head_Wd"], dhin, Mh, c3, 27 * cmid, Zo * cmid, 27 * cmid, D7 * c3, dtype=dt,
                     a_mode=L.A_CONV3, gh=H2, gw=W2, cs=3 * cmid, nz=D7, a_coff=[z * cmid for z in zs],
                     b_off=[zp * c3 * 27 * cmid for zp in range(D7)], c_coff=[zp * c3 for zp in range(D7)])
        del dU
        d = o.head_shuffle_bwd(dhin, B, fh, fw, c3, D7, cfg["head_pool"])
        del dhin
        return d

    # ------------------------------------------------------------------ backward
    def backward(self, sv, dout: Tensor) -> None:
        """Accumulates parameter gradients into the flat gradient buffer (no input gradient:
        the image stack never requires grad on this path).  ``on_bucket_ready(i)`` fires as bucket i completes — but only
        during the LAST outstanding backward of the step: with several forwards per step (CombinedLoader batches, DynaCLR
        with unpaired forwards) an earlier backward leaves buckets that later ones still accumulate into."""
        last = self._pending_bwd <= 1
        self._pending_bwd = max(self._pending_bwd - 1, 0)
        for i in self.backward_stages(sv, dout):
            if last and self.on_bucket_ready:
                self.on_bucket_ready(i)

    def backward_stages(self, sv, dout: Tensor):
        """Generator form of the backward schedule: yields the bucket index (0 = head + decoder, 1 = encoder stages 3-2,
        2 = stages 1-0 + stem) right after the last launch that writes into that bucket of the flat gradient buffer.
        ``viscy_amd.step.TrainStep`` captures the stretch between two yields as one hipGraph segment and issues the
        bucket's RCCL all-reduce between the segments."""
        # the gradient finalisers of a segment (unprep_grad / matvec_t_add / accumulating transposes: ~100 launches of 5 - 20 us
        # that only touch weight-sized data) are collected and launched as task lists when the segment ends (ops.batch_open)
        self.ops.batch_open()
        try:
            o, cfg, m, g, W = self.ops, self.cfg, self.model, self.g, sv["W"]
            dt = sv["dt"]
            B, H, Wd = sv["shape"]
            dev = self.device
            za_key = ("bwd", B, H, Wd, sv["masked"])
            self._za = za = _ZeroArena(dev, self._za_need.get(za_key, 0), self.ops)
            if cfg.get("head") == "embed":
                d = self._embed_tail_bwd(sv, dout[0], dout[1], dt, B)
            elif cfg.get("head", "conv") == "shuffle":
                fh, fw = sv["head"]
                d = o.voxel_shuffle_bwd(dout.contiguous().float(), B, fh, fw, cfg["out_channels"], cfg["out_stack_depth"],
                                        cfg["stem_kernel"][-1], True, dt)
            else:
                hdt = self._head_dtype(dt)
                d = self._head_conv_bwd(sv, dout, hdt, B, dev)
                if hdt != dt:
                    d = d.to(dt)
            # ---- decoder (reverse)
            dskips = {}
            for k in (2, 1, 0) if cfg.get("head") != "embed" else ():
                proj, blocks = W["dec"][k]
                st_sv = sv["dec"][k]
                cat, xn, mean, rstd, c_up, sc = st_sv["proj"]
                _, sh, sw_, _ = (None, *sv["feat_dims"][2 - k])
                Mk, ccat = B * sh * sw_, c_up + sc
                for bw, bsv in zip(reversed(blocks), reversed(st_sv["blocks"])):
                    d = self._block_bwd(d, bw, bsv, B, sh, sw_, dt)
                o.gemm("tn", xn, d, g(proj.conv.weight), Mk, proj.cout, ccat, ccat, proj.cout, ccat, dtype=dt,
                       colsum=g(proj.conv.bias))
                dxn = torch.empty((Mk, ccat), dtype=dt, device=dev)
                o.gemm("nt", d, proj.WT, dxn, Mk, ccat, proj.cout, proj.cout, proj.cout, ccat, dtype=dt)
                dcat = o.ln_bwd(dxn, cat, mean, rstd, proj.ln.weight, None, g(proj.ln.weight), g(proj.ln.bias), Mk, ccat)
                del dxn
                d, dskips[2 - k] = o.pixel_shuffle_cat_bwd(dcat, B, sh // 2, sw_ // 2, c_up, sc)
                del dcat
                pre = W["dec_pre"][k]
                if pre is not None:  # pre-convolution: weight gradient from the re-gathered patch matrix, data gradient through its transpose
                    pc, _, WpT = pre
                    lh, lw, cpc = sh // 2, sw_ // 2, 4 * c_up
                    Ml = B * lh * lw
                    col = o.im2col3x3(st_sv["pre_in"], B, lh, lw, cpc)
                    dWp = self._za.take(cpc, 9 * cpc)
                    o.gemm("tn", col, d, dWp, Ml, cpc, 9 * cpc, 9 * cpc, cpc, 9 * cpc, dtype=dt, colsum=g(pc.bias))
                    o.unprep_grad(dWp, g(pc.weight), cpc, cpc, 9)
                    del col
                    dcol = torch.empty((Ml, 9 * cpc), dtype=dt, device=dev)
                    o.gemm("nt", d, WpT, dcol, Ml, 9 * cpc, cpc, cpc, cpc, 9 * cpc, dtype=dt)
                    d = o.col2im3x3(dcol, B, lh, lw, cpc)
                    del dcol
            yield from self._segment_end(0)
            if self.encoder_frozen():  # nothing below the decoder needs a gradient: skip ~40 % of the backward
                self._za_need[za_key] = za.used
                yield from self._segment_end(1)
                yield from self._segment_end(2)
                return
            # ---- encoder (reverse); d = gradient w.r.t. feats[3]
            for i in (3, 2, 1, 0):
                proj, blocks = W["enc"][i]
                st_sv = sv["enc"][i]
                ch, cw, cc = sv["feat_dims"][i]
                for bw, bsv in zip(reversed(blocks), reversed(st_sv["blocks"])):
                    d = self._block_bwd(d, bw, bsv, B, ch, cw, dt)
                if proj is not None:
                    prev, xn, mean, rstd = st_sv["proj"]
                    cin = proj.cin
                    Mo = B * ch * cw
                    dWg = self._za.take(proj.cout, 4 * cin)
                    o.gemm("tn", xn, d, dWg, Mo, proj.cout, 4 * cin, cin, proj.cout, 4 * cin, dtype=dt, a_mode=L.A_PATCH2,
                           gh=ch, gw=cw, cs=cin, colsum=g(proj.conv.bias))
                    o.unprep_grad(dWg, g(proj.conv.weight), proj.cout, cin, 4)
                    dxn = torch.empty((B * 4 * ch * cw, cin), dtype=dt, device=dev)
                    o.gemm("nt", d, proj.WT, dxn, Mo, 4 * cin, proj.cout, proj.cout, proj.cout, cin, dtype=dt,
                           c_mode=L.A_PATCH2, c_cs=cin, gh=ch, gw=cw)
                    d = o.ln_bwd(dxn, prev, mean, rstd, proj.ln.weight, dskips.get(i - 1), g(proj.ln.weight), g(proj.ln.bias),
                                 B * 4 * ch * cw, cin)
                    del dxn
                if i == 2:
                    yield from self._segment_end(1)
            # ---- stem_1 LayerNorm + stem projection
            P, f, mean, rstd = sv["stem"]
            ln1 = m.encoder_stages.stem_1
            M0, C0, K0 = f.shape[0], f.shape[1], P.shape[1]
            df = o.ln_bwd(d, f, mean, rstd, ln1.weight, None, g(ln1.weight), g(ln1.bias), M0, C0)
            Dp = cfg["ratio"]
            if sv.get("flat_stem"):
                Kw = m.stem2d.weight[0].numel()
                o.gemm("tn", P, df, g(m.stem2d.weight), M0, C0, Kw, K0, C0, Kw, dtype=dt, colsum=g(m.stem2d.bias))
            elif Dp == 1:
                Kw = m.stem.conv.weight[0].numel()  # the patch matrix may carry zero-padded tail columns (lda = K0 >= Kw)
                o.gemm("tn", P, df, g(m.stem.conv.weight), M0, C0, Kw, K0, C0, Kw, dtype=dt, colsum=g(m.stem.conv.bias))
            else:
                co3 = C0 // Dp
                K = K0 // Dp
                dWe = self._za.take(C0, K0)
                dbe = self._za.take(C0)
                o.gemm("tn", P, df, dWe, M0, C0, K0, K0, C0, K0, dtype=dt, colsum=dbe)
                dWe = dWe.view(co3, Dp, Dp, K)
                g(m.stem.conv.weight).add_(torch.stack([dWe[:, dd, dd] for dd in range(Dp)], 0).sum(0).view_as(m.stem.conv.weight))
                g(m.stem.conv.bias).add_(dbe.view(co3, Dp).sum(1))
            self._za_need[za_key] = za.used
            yield from self._segment_end(2)


        finally:
            self.ops.batch_close()

    def _segment_end(self, k: int):
        self.ops.batch_close()
        yield k
        self.ops.batch_open()

# ------------------------------------------------------------------------------------------------
class _UNeXt2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, dt, need_bwd, masks, bn_groups, *params):
        eng = model.engine()
        out, sv = eng.forward(x, dt, need_bwd, masks, bn_groups)
        ctx.model, ctx.sv = model, sv
        return out

    @staticmethod
    def backward(ctx, *douts):
        dout = douts[0] if len(douts) == 1 else douts  # the embedding tail returns (embedding, projection)
        model, sv = ctx.model, ctx.sv
        if sv is None:
            raise RuntimeError("viscy_amd.UNeXt2: backward called but the forward ran without gradient bookkeeping")
        eng = model.engine()
        ctx.sv = None
        if model.grad_mode == "flat":
            eng.backward(sv, dout)
            return (None, None, None, None, None, None) + tuple(None for _ in eng.order)
        # autograd mode: compute into a zeroed flat buffer and hand views back to autograd
        saved = eng.flat_grad
        eng.flat_grad = torch.zeros_like(saved)
        old = eng.grad_of
        eng.grad_of = {}
        for p, off in zip(eng.order, eng.offsets):
            eng.grad_of[id(p)] = eng.flat_grad[off : off + p.numel()].view(p.shape)
        try:
            eng.backward(sv, dout)
            grads = tuple(eng.grad_of[id(p)] if p.requires_grad else None for p in eng.order)
        finally:
            eng.flat_grad, eng.grad_of = saved, old
        return (None, None, None, None, None, None) + grads


def unext2_apply(model, x: Tensor, masks=None, bn_groups: int = 1) -> Tensor:
    eng = model.engine()
    dt = model._resolve_dtype()
    if model.grad_mode == "flat":
        eng.attach_grads()
    need_bwd = torch.is_grad_enabled() and any(p.requires_grad for p in eng.order)
    with torch.autocast("cuda", enabled=False):
        return _UNeXt2Fn.apply(x.float(), model, dt, need_bwd, masks, bn_groups, *eng.order)
