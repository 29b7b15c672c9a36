"""Decoder: drop-in for the reference's ``rnn.RNN`` (rnn.py:10-108) and, through the
``cell`` switch, ``LSTM/rnn_lstm.RNN`` (rnn_lstm.py:8-57).

Same constructor, same attribute names (``embeddings``, ``unit``, ``linear``) and
``state_dict`` keys, same call conventions:

    logits = rnn(cnn_feature, image_caption, caption_size)      # (N_tok, V), time-major packed rows
    ids    = rnn.sentence_index(cnn_feature, beam_size=0)       # Long(B, 25) (squeezed)

``nn.Embedding`` / ``nn.GRU`` / ``nn.Linear`` objects are parameter containers only; all
arithmetic runs in libshowtell_hip (st_rnn_forward / st_rnn_backward / st_rnn_greedy /
st_cross_entropy).  ``loss()`` is the fused route (vocabulary projection + cross entropy
without handing fp32 logits to torch) used by the training driver and bench.py.

Gradients: the backward kernels accumulate straight into ``param.grad`` (fp32), so after
``loss.backward()`` the optimizer sees exactly what torch autograd would have produced.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ST_BF16, ST_CELL_GRU, ST_CELL_LSTM, ST_F32, RnnGrads, RnnParams, check, lib
from .seq import plan_for

CAP_MAX = 25  # rnn.py:39


def _cp(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def working_copy(p, dtype):
    """The tensor the kernels read for parameter `p`: itself in fp32 mode, a cached bf16
    shadow otherwise (refreshed when the parameter was modified or moved)."""
    if dtype == torch.float32:
        return p.data
    sh = getattr(p, "_st_shadow", None)
    if (sh is None or sh.device != p.device or getattr(p, "_st_shadow_ver", -1) != p._version
            or getattr(p, "_st_shadow_ptr", 0) != p.data_ptr()):
        if sh is None or sh.device != p.device or sh.shape != p.shape:
            sh = torch.empty(p.shape, device=p.device, dtype=dtype)
        ops.cast(p.data.contiguous(), dtype, out=sh)
        p._st_shadow, p._st_shadow_ver, p._st_shadow_ptr = sh, p._version, p.data_ptr()
    return sh


def grad_buffer(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p.data, dtype=torch.float32)
    return p.grad


def up8(v):
    return (v + 7) // 8 * 8


class _DecoderFn(torch.autograd.Function):
    """mode 'logits': returns fp32 logits rows; mode 'loss': returns the mean cross entropy."""

    @staticmethod
    def forward(ctx, feat, _anchor, module, caption, lens, mode, need_grad):
        m = module
        dev = feat.device
        if not feat.is_cuda or not m.linear.weight.is_cuda:
            raise _lib.ShowTellHipError("the decoder and its inputs must live on a HIP device (no CPU fallback in the MI355X build)")
        plan = plan_for(lens, dev)
        caption = caption.contiguous()
        seq = plan.c_struct(caption)
        prm, keep = m._c_params()
        dt = m.compute_dtype
        featd = feat.detach().contiguous()
        featd = featd if featd.dtype == dt else ops.cast(featd.float(), dt)
        nbytes = lib().st_rnn_workspace_bytes(C.byref(prm), C.byref(seq))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        V, Vp, n = m.vocab_size, lib().st_rnn_vocab_ld(m.vocab_size), plan.ntok
        targets = torch.empty(n, device=dev, dtype=torch.long)
        # mode 'loss' in bf16: the vocabulary projection and the cross entropy run tile by tile without a logits tensor (csrc/vocab_ce.hip);
        # ST_FUSED_CE=0 keeps st_rnn_forward's logits + st_cross_entropy
        fused = mode == "loss" and os.environ.get("ST_FUSED_CE", "1") != "0" and bool(lib().st_rnn_fused_loss_supported(C.byref(prm)))
        logits = None if fused else torch.empty(n, Vp, device=dev, dtype=torch.float32 if mode == "logits" else dt)
        check(lib().st_rnn_forward(C.byref(prm), C.byref(seq), None, _cp(featd), _cp(ws), nbytes, _cp(logits),
                                   ST_F32 if (logits is not None and logits.dtype == torch.float32) else ST_BF16, Vp, _cp(targets),
                                   int(need_grad), _stream()), "st_rnn_forward")
        ctx.m, ctx.plan, ctx.caption, ctx.ws, ctx.mode, ctx.keep = m, plan, caption, ws, mode, keep
        ctx.feat_dtype = feat.dtype
        ctx.fused = fused
        if mode == "logits":
            return logits[:, :V]
        loss = torch.zeros((), device=dev, dtype=torch.float32)
        if fused:
            sb = lib().st_rnn_fused_loss_bytes(C.byref(prm), C.byref(seq))
            scratch = torch.empty(sb // 4, device=dev, dtype=torch.float32)
            check(lib().st_rnn_fused_loss(C.byref(prm), C.byref(seq), _cp(ws), nbytes, _cp(targets), _cp(scratch), sb, _cp(loss), _stream()),
                  "st_rnn_fused_loss")
            ctx.logits, ctx.targets, ctx.scratch = None, targets, scratch
            return loss
        check(lib().st_cross_entropy(_cp(logits), ST_F32 if dt == torch.float32 else ST_BF16, _cp(targets), n, V, Vp,
                                     _cp(loss), None, 0, Vp, 1.0, None, _stream()), "st_cross_entropy")
        ctx.logits, ctx.targets = logits, targets
        return loss

    @staticmethod
    def backward(ctx, gout):
        m, plan = ctx.m, ctx.plan
        dev = gout.device
        dt = m.compute_dtype
        dtc = ST_F32 if dt == torch.float32 else ST_BF16
        V, Vp, n = m.vocab_size, lib().st_rnn_vocab_ld(m.vocab_size), plan.ntok
        if ctx.mode == "logits":
            g = gout if gout.dtype == torch.float32 else gout.float()
            dlog = torch.zeros(n, Vp, device=dev, dtype=dt)   # pad columns feed the backward GEMM as K: must be zero
            gs = g.stride(0) if g.stride(1) == 1 else None
            if gs is None:
                g = g.contiguous(); gs = g.stride(0)
            check(lib().st_cast2d(_cp(g), _cp(dlog), ST_F32, dtc, n, V, gs, Vp, _stream()), "st_cast2d")
        elif ctx.fused:
            gsc = gout.detach().float().contiguous()
            dlog = torch.empty(n, Vp, device=dev, dtype=dt)
            prm_, keep_ = m._c_params()
            seq_ = plan.c_struct(ctx.caption)
            check(lib().st_rnn_fused_dlogits(C.byref(prm_), C.byref(seq_), _cp(ctx.ws), ctx.ws.numel(), _cp(ctx.targets), _cp(ctx.scratch),
                                             _cp(gsc), _cp(dlog), Vp, _stream()), "st_rnn_fused_dlogits")
        else:
            dlog = ctx.logits   # overwritten in place by (softmax - onehot) * dLoss / N_tok
            gsc = gout.detach().float().contiguous()
            check(lib().st_cross_entropy(_cp(ctx.logits), dtc, _cp(ctx.targets), n, V, Vp, None, _cp(dlog), dtc, Vp, 1.0,
                                         _cp(gsc), _stream()), "st_cross_entropy(bwd)")
        prm, keep = m._c_params()
        grads, keep2 = m._c_grads()
        seq = plan.c_struct(ctx.caption)
        dfeat = torch.empty(plan.B, m.embed_dim, device=dev, dtype=torch.float32)
        check(lib().st_rnn_backward(C.byref(prm), C.byref(grads), C.byref(seq), None, _cp(dlog), Vp, None, _cp(ctx.ws),
                                    ctx.ws.numel(), _cp(dfeat), None, _stream()), "st_rnn_backward")
        ctx.ws = None
        return dfeat.to(ctx.feat_dtype), None, None, None, None, None, None


class RNN(torch.nn.Module):

    cell = "gru"

    def __init__(self, embed_dim, num_hidden_units, vocab_size, num_layers, dtype=torch.float32):
        '''
        Args (as the reference, rnn.py:12-19):
            embed_dim (int) : Embedding dimension between CNN and RNN
            num_hidden_units (int) : Number of hidden units
            vocab_size (int) : Size of the vocabulary
            num_layers (int) : # of layers
            dtype : kernel storage type (float32 = parity mode, bfloat16 = performance mode)
        '''
        super(RNN, self).__init__()
        if num_layers > _lib.ST_MAX_LAYERS:
            raise ValueError(f"num_layers={num_layers} exceeds the kernel limit of {_lib.ST_MAX_LAYERS}")
        self.embeddings = nn.Embedding(vocab_size, embed_dim)
        unit_cls = nn.GRU if self.cell == "gru" else nn.LSTM
        self.unit = unit_cls(embed_dim, num_hidden_units, num_layers, batch_first=True)
        self.linear = nn.Linear(num_hidden_units, vocab_size)
        self.embed_dim, self.hidden, self.vocab_size, self.num_layers = embed_dim, num_hidden_units, vocab_size, num_layers
        self.compute_dtype = dtype

    # ---- C descriptors ---------------------------------------------------------------
    def _layer_params(self):
        return [(getattr(self.unit, f"weight_ih_l{l}"), getattr(self.unit, f"weight_hh_l{l}"),
                 getattr(self.unit, f"bias_ih_l{l}"), getattr(self.unit, f"bias_hh_l{l}")) for l in range(self.num_layers)]

    def _c_params(self):
        dt = self.compute_dtype
        if not self.linear.weight.is_cuda:
            raise _lib.ShowTellHipError("the decoder must live on a HIP device (no CPU fallback in the MI355X build)")
        p = RnnParams()
        p.cell = ST_CELL_GRU if self.cell == "gru" else ST_CELL_LSTM
        p.dtype = ST_F32 if dt == torch.float32 else ST_BF16
        p.L, p.in0, p.H, p.V, p.E = self.num_layers, self.embed_dim, self.hidden, self.vocab_size, self.embed_dim
        keep = []
        e = working_copy(self.embeddings.weight, dt); keep.append(e); p.emb = e.data_ptr()
        for l, (wi, wh, bi, bh) in enumerate(self._layer_params()):
            a, b = working_copy(wi, dt), working_copy(wh, dt)
            keep += [a, b]
            p.w_ih[l], p.w_hh[l], p.b_ih[l], p.b_hh[l] = a.data_ptr(), b.data_ptr(), bi.data.data_ptr(), bh.data.data_ptr()
        w = working_copy(self.linear.weight, dt); keep.append(w)
        p.w_lin, p.b_lin = w.data_ptr(), self.linear.bias.data.data_ptr()
        return p, keep

    def _c_grads(self):
        g = RnnGrads()
        keep = []
        def gb(prm):
            t = grad_buffer(prm); keep.append(t); return t.data_ptr()
        g.emb = gb(self.embeddings.weight)
        for l, (wi, wh, bi, bh) in enumerate(self._layer_params()):
            g.w_ih[l], g.w_hh[l], g.b_ih[l], g.b_hh[l] = gb(wi), gb(wh), gb(bi), gb(bh)
        g.w_lin, g.b_lin = gb(self.linear.weight), gb(self.linear.bias)
        return g, keep

    # ---- reference surface -------------------------------------------------------------
    def forward(self, cnn_feature, image_caption, caption_size):
        """rnn.py:27-35: teacher-forced logits over the packed sequence, (N_tok, V) fp32."""
        return _DecoderFn.apply(cnn_feature, self.linear.bias, self, image_caption, caption_size, "logits", torch.is_grad_enabled())

    def loss(self, cnn_feature, image_caption, caption_size):
        """main.py:145-149 fused: CrossEntropyLoss()(rnn(feat, cap, lens), packed(cap)) as one scalar."""
        return _DecoderFn.apply(cnn_feature, self.linear.bias, self, image_caption, caption_size, "loss", torch.is_grad_enabled())

    def beam_search(self, cnn_feature, beam_width=4, num_hypotheses=1, max_length=50, start_id=1, end_id=2):
        """beam_search.py:45-97 over the whole batch (BASELINE config 5: beam_width=5, max_length=25)."""
        from .beam import beam_search
        return beam_search(self, cnn_feature, beam_width, num_hypotheses, max_length, start_id, end_id)

    def sentence_index(self, cnn_feature, beam_size=0, return_logits=False):
        """rnn.py:37-108: 25-step greedy decode (beam_size=0) or the bs=1 ranking loop (beam_size>0)."""
        if beam_size and beam_size > 0:
            from .beam import quirky_beam
            return quirky_beam(self, cnn_feature, beam_size)
        with torch.no_grad():
            prm, keep = self._c_params()
            dt = self.compute_dtype
            feat = cnn_feature.detach().contiguous()
            feat = feat if feat.dtype == dt else ops.cast(feat.float(), dt)
            B = feat.shape[0]
            nbytes = lib().st_rnn_greedy_workspace_bytes(C.byref(prm), B)
            ws = torch.empty(nbytes, device=feat.device, dtype=torch.uint8)
            ids = torch.empty(B, CAP_MAX, device=feat.device, dtype=torch.long)
            lg = torch.empty(CAP_MAX, B, up8(self.vocab_size), device=feat.device) if return_logits else None
            check(lib().st_rnn_greedy(C.byref(prm), _cp(feat), B, CAP_MAX, _cp(ws), nbytes, _cp(ids), _cp(lg), _stream()),
                  "st_rnn_greedy")
        out = ids.squeeze()                                               # rnn.py:56
        if return_logits:
            return out, lg[:, :, :self.vocab_size].permute(1, 0, 2)
        return out
