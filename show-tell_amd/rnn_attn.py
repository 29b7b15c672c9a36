"""Soft-attention decoder: drop-in for ``Attention/rnn_attn.py`` (``Attention_Net``, ``RNN_Attn``) and,
through the ``cell`` switch, ``Attention/rnn_attn_LSTM.py``.

    logits, alphas = rnn(cnn_feature, image_caption, caption_size)    # (N_tok, V) packed rows, (B, T, P) zero padded
    ids            = rnn.sentence_index(cnn_feature, vocab)           # Long(B, 25)

Same constructor, attribute names (``embeddings, unit, linear, init_h, attn.{encoder_att, decoder_att,
full_att}, embed`` [+ ``init_c``]) and ``state_dict`` keys as the reference; the sub-modules are parameter
containers, the arithmetic runs in st_attn_forward / st_attn_backward / st_attn_greedy.

Reference quirks kept: the input token at step t is ``caption[:, t]`` (the same token
that is the target of that step), the initial hidden state is replicated over all layers, attention is keyed
on the last layer's state, ``alphas`` stay zero beyond each caption's length, unlike the reference the
module does not hard-code ``.cuda()`` but still requires a HIP device.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ST_BF16, ST_CELL_GRU, ST_CELL_LSTM, ST_F32, AttnGrads, AttnParams, check, lib
from .rnn import CAP_MAX, _cp, _stream, grad_buffer, up8, working_copy
from .seq import plan_for


class Attention_Net(nn.Module):
    """Parameter container with the reference's attribute names (rnn_attn.py:13-19)."""

    def __init__(self, nos_filters, num_hidden_units, attention_dim=512):
        super(Attention_Net, self).__init__()
        self.encoder_att = nn.Linear(nos_filters, attention_dim)
        self.decoder_att = nn.Linear(num_hidden_units, attention_dim)
        self.full_att = nn.Linear(attention_dim, 1)
        self.relu = nn.LeakyReLU(negative_slope=0.2)
        self.softmax = nn.Softmax(dim=1)

    def forward(self, img_feat, hidden_state):  # pragma: no cover - container
        raise _lib.ShowTellHipError("Attention_Net is a parameter container; call RNN_Attn.forward")


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, _anchor, module, caption, lens, mode, alpha_c, need_grad):
        m = module
        dev = feat.device
        if not feat.is_cuda or not m.linear.weight.is_cuda:
            raise _lib.ShowTellHipError("the attention decoder and its inputs must live on a HIP device (no CPU fallback)")
        plan = plan_for(lens, dev)
        caption = caption.contiguous()
        seq = plan.c_struct(caption)
        cap_T = caption.t().contiguous()
        prm, keep = m._c_params()
        dt = m.compute_dtype
        dtc = ST_F32 if dt == torch.float32 else ST_BF16
        B, Fd, P = feat.shape
        if Fd != m.nos_filters or P != m.num_pixels_checked(P):
            raise _lib.ShowTellHipError(f"cnn_feature must be (B, {m.nos_filters}, P), got {tuple(feat.shape)}")
        prm.P = P
        featc = feat.detach().float().contiguous()
        nbytes = lib().st_attn_workspace_bytes(C.byref(prm), C.byref(seq))
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        V, Vp, n, T = m.vocab_size, up8(m.vocab_size), plan.ntok, caption.shape[1]
        alphas = torch.zeros(B, T, P, device=dev, dtype=torch.float32)        # rnn_attn.py:65
        logits = torch.empty(n, Vp, device=dev, dtype=torch.float32 if mode == "logits" else dt)
        check(lib().st_attn_forward(C.byref(prm), C.byref(seq), _cp(featc), _cp(cap_T), _cp(ws), nbytes, _cp(logits),
                                    ST_F32 if logits.dtype == torch.float32 else ST_BF16, Vp, _cp(alphas), int(need_grad), _stream()),
              "st_attn_forward")
        # st_attn_forward steps over plan.T columns; alphas has T (padded) columns: strides must agree
        ctx.m, ctx.plan, ctx.caption, ctx.cap_T, ctx.ws, ctx.mode, ctx.P = m, plan, caption, cap_T, ws, mode, P
        ctx.alphas, ctx.alpha_c, ctx.keep = alphas, alpha_c, keep
        if mode == "logits":
            return logits[:, :V], alphas
        targets = torch.nn.utils.rnn.pack_padded_sequence(caption, lens, batch_first=True)[0].contiguous()   # main_attn.py:126
        loss = torch.zeros((), device=dev, dtype=torch.float32)
        check(lib().st_cross_entropy(_cp(logits), dtc, _cp(targets), n, V, Vp, _cp(loss), None, 0, Vp, 1.0, None, _stream()), "st_cross_entropy")
        check(lib().st_attn_reg_loss(_cp(alphas), B, T, P, float(alpha_c), _cp(loss), _stream()), "st_attn_reg_loss")
        ctx.logits, ctx.targets = logits, targets
        return loss, alphas

    @staticmethod
    def backward(ctx, g0, galphas):
        m, plan = ctx.m, ctx.plan
        dev = g0.device
        dt = m.compute_dtype
        dtc = ST_F32 if dt == torch.float32 else ST_BF16
        V, Vp, n = m.vocab_size, up8(m.vocab_size), plan.ntok
        gs = None
        dal = None
        if ctx.mode == "logits":
            g = g0 if g0.dtype == torch.float32 else g0.float()
            if g.stride(1) != 1:
                g = g.contiguous()
            dlog = torch.empty(n, Vp, device=dev, dtype=dt)
            check(lib().st_cast2d(_cp(g), _cp(dlog), ST_F32, dtc, n, V, g.stride(0), Vp, _stream()), "st_cast2d")
            dal = (galphas if galphas is not None else torch.zeros_like(ctx.alphas)).float().contiguous()
            alpha_c = 0.0
        else:
            dlog = ctx.logits
            gs = g0.detach().float().contiguous()
            check(lib().st_cross_entropy(_cp(ctx.logits), dtc, _cp(ctx.targets), n, V, Vp, None, _cp(dlog), dtc, Vp, 1.0, _cp(gs), _stream()),
                  "st_cross_entropy(bwd)")
            alpha_c = ctx.alpha_c
        prm, keep = m._c_params()
        prm.P = ctx.P
        grads, keep2 = m._c_grads()
        seq = plan.c_struct(ctx.caption)
        check(lib().st_attn_backward(C.byref(prm), C.byref(grads), C.byref(seq), _cp(ctx.cap_T), _cp(dlog), Vp, _cp(ctx.alphas), _cp(dal),
                                     float(alpha_c), _cp(gs), _cp(ctx.ws), ctx.ws.numel(), _stream()), "st_attn_backward")
        ctx.ws = None
        return None, None, None, None, None, None, None, None


class RNN_Attn(nn.Module):

    cell = "gru"

    def __init__(self, embed_dim, nos_filters, attention_dim, num_hidden_units, vocab_size, num_layers, dtype=torch.float32):
        '''
        Args (as the reference, rnn_attn.py:35-42) + dtype (kernel storage type).
        '''
        super(RNN_Attn, self).__init__()
        if num_layers > _lib.ST_MAX_LAYERS:
            raise ValueError(f"num_layers={num_layers} exceeds the kernel limit of {_lib.ST_MAX_LAYERS}")
        self.nos_filters = nos_filters
        self.num_layers = num_layers
        self.vocab_size = vocab_size
        self.embeddings = nn.Embedding(vocab_size, embed_dim)
        unit_cls = nn.GRU if self.cell == "gru" else nn.LSTM
        self.unit = unit_cls(2 * embed_dim, num_hidden_units, num_layers, batch_first=True)
        self.linear = nn.Linear(num_hidden_units, vocab_size)
        self.cap_max_size = CAP_MAX                                          # rnn_attn.py:53
        self.init_h = nn.Linear(nos_filters, num_hidden_units)
        if self.cell != "gru":
            self.init_c = nn.Linear(nos_filters, num_hidden_units)           # rnn_attn_LSTM.py:55
        self.attn = Attention_Net(nos_filters, num_hidden_units, attention_dim)
        self.embed = nn.Linear(nos_filters, embed_dim)
        self.embed_dim, self.hidden, self.attention_dim = embed_dim, num_hidden_units, attention_dim
        self.compute_dtype = dtype

    def num_pixels_checked(self, P):
        if P > 64:
            raise _lib.ShowTellHipError(f"at most 64 feature-map pixels are supported (got {P})")
        return P

    def _layer_params(self):
        return [(getattr(self.unit, f"weight_ih_l{l}"), getattr(self.unit, f"weight_hh_l{l}"),
                 getattr(self.unit, f"bias_ih_l{l}"), getattr(self.unit, f"bias_hh_l{l}")) for l in range(self.num_layers)]

    def _c_params(self):
        dt = self.compute_dtype
        p = AttnParams()
        r = p.rnn
        r.cell = ST_CELL_GRU if self.cell == "gru" else ST_CELL_LSTM
        r.dtype = ST_F32 if dt == torch.float32 else ST_BF16
        r.L, r.in0, r.H, r.V, r.E = self.num_layers, 2 * self.embed_dim, self.hidden, self.vocab_size, self.embed_dim
        keep = []
        def wc(prm):
            t = working_copy(prm, dt); keep.append(t); return t.data_ptr()
        r.emb = wc(self.embeddings.weight)
        for l, (wi, wh, bi, bh) in enumerate(self._layer_params()):
            r.w_ih[l], r.w_hh[l], r.b_ih[l], r.b_hh[l] = wc(wi), wc(wh), bi.data.data_ptr(), bh.data.data_ptr()
        r.w_lin, r.b_lin = wc(self.linear.weight), self.linear.bias.data.data_ptr()
        p.F, p.A, p.P = self.nos_filters, self.attention_dim, 49
        a = self.attn
        p.w_enc, p.b_enc = wc(a.encoder_att.weight), a.encoder_att.bias.data.data_ptr()
        p.w_dec, p.b_dec = wc(a.decoder_att.weight), a.decoder_att.bias.data.data_ptr()
        p.w_full, p.b_full = a.full_att.weight.data.data_ptr(), a.full_att.bias.data.data_ptr()
        p.w_init_h, p.b_init_h = wc(self.init_h.weight), self.init_h.bias.data.data_ptr()
        if self.cell != "gru":
            p.w_init_c, p.b_init_c = wc(self.init_c.weight), self.init_c.bias.data.data_ptr()
        p.w_embed, p.b_embed = wc(self.embed.weight), self.embed.bias.data.data_ptr()
        return p, keep

    def _c_grads(self):
        g = AttnGrads()
        keep = []
        def gb(prm):
            t = grad_buffer(prm); keep.append(t); return t.data_ptr()
        g.rnn.emb = gb(self.embeddings.weight)
        for l, (wi, wh, bi, bh) in enumerate(self._layer_params()):
            g.rnn.w_ih[l], g.rnn.w_hh[l], g.rnn.b_ih[l], g.rnn.b_hh[l] = gb(wi), gb(wh), gb(bi), gb(bh)
        g.rnn.w_lin, g.rnn.b_lin = gb(self.linear.weight), gb(self.linear.bias)
        a = self.attn
        g.w_enc, g.b_enc = gb(a.encoder_att.weight), gb(a.encoder_att.bias)
        g.w_dec, g.b_dec = gb(a.decoder_att.weight), gb(a.decoder_att.bias)
        g.w_full, g.b_full = gb(a.full_att.weight), gb(a.full_att.bias)
        g.w_init_h, g.b_init_h = gb(self.init_h.weight), gb(self.init_h.bias)
        if self.cell != "gru":
            g.w_init_c, g.b_init_c = gb(self.init_c.weight), gb(self.init_c.bias)
        g.w_embed, g.b_embed = gb(self.embed.weight), gb(self.embed.bias)
        return g, keep

    def forward(self, cnn_feature, image_caption, caption_size):
        """rnn_attn.py:98-118: (packed logits (N_tok,V) fp32, alphas (B,T,P))."""
        return _AttnFn.apply(cnn_feature, self.linear.bias, self, image_caption, caption_size, "logits", 0.0, torch.is_grad_enabled())

    def loss(self, cnn_feature, image_caption, caption_size, alpha_c=1.0):
        """main_attn.py:126-131 fused: CE(packed logits, packed caption) + alpha_c * mean((1 - sum_t alpha)^2)."""
        out, _ = _AttnFn.apply(cnn_feature, self.linear.bias, self, image_caption, caption_size, "loss", float(alpha_c), torch.is_grad_enabled())
        return out

    def sentence_index(self, cnn_feature, vocab):
        """rnn_attn.py:120-145: greedy decode from vocab('<start>'), exactly cap_max_size steps."""
        ind = vocab('<start>')                                               # rnn_attn.py:127
        with torch.no_grad():
            if not cnn_feature.is_cuda:
                raise _lib.ShowTellHipError("cnn_feature must be on the HIP device (no CPU fallback)")
            prm, keep = self._c_params()
            B, Fd, P = cnn_feature.shape
            prm.P = self.num_pixels_checked(P)
            featc = cnn_feature.detach().float().contiguous()
            nbytes = lib().st_attn_greedy_workspace_bytes(C.byref(prm), B)
            ws = torch.empty(nbytes, device=featc.device, dtype=torch.uint8)
            ids = torch.empty(B, self.cap_max_size, device=featc.device, dtype=torch.long)
            check(lib().st_attn_greedy(C.byref(prm), _cp(featc), B, self.cap_max_size, int(ind), _cp(ws), nbytes, _cp(ids), _stream()),
                  "st_attn_greedy")
        return ids.squeeze()                                                 # rnn_attn.py:143
