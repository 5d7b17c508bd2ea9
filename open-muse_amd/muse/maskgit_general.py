"""The general form of muse.MaskGitTransformer: everything the reference class builds that the flat class-conditional engine
(modeling_transformer.py: layernorm + NormFormer + MLM head, no text) does not -

  * text conditioning: `add_cross_attention` (a cross-attention block per layer, reference muse/modeling_transformer.py:886-899),
    `project_encoder_hidden_states` (`encoder_proj` + `encoder_proj_layer_norm`, :1143-1146, :1239-1241), condition dropout for
    classifier-free guidance (`cond_dropout_prob`, :1243-1247) and guided decoding in `generate2` (:1394-1416);
  * `norm_type="rmsnorm"` (:75-100; every norm but the feed-forward's `pre_mlp_layer_norm`, which the reference always builds as a
    LayerNorm, :768-770), `use_normformer=False`, `use_encoder_layernorm=False`, `use_mlm_layer=False` / `use_mlm_layernorm=False`.

This is what `training/train_muse.py:736-750` drives when the model class is MaskGitTransformer (configs/cc12m.yaml,
configs/imagenet_text2image.yaml: hidden 1024, 24 layers, rmsnorm, no NormFormer, T5 states of width 1024).

Parameters are ordinary tensors named like the reference's state dict; the forward records a tape, the backward is hand-written
over the same libmuse_hip kernels as muse.MaskGiTUViT_v2 (tape_ops.TapeOps): bf16 mode = fused self / cross attention at any
length, bf16 GEMM operands with f32 accumulation, f32 residual stream / norms / loss; f32 mode = the reference's algorithm with
exact-f32 MFMA products.  muse.FusedAdamW steps such a model with one multi-tensor launch, muse.GradReducer all-reduces it in its
tensor-list mode.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from ._hip import MuseHipError
from .tape_ops import TapeOps


class _P(nn.Module):
    """leaf named like the reference's nn.Linear / nn.Embedding / norm module (`.weight`)"""

    def __init__(self, *shape, ones=False, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(*shape) if ones else torch.empty(*shape))
        if bias:                     # `use_bias`: nn.Linear / LayerNorm bias, zero-initialised (reference :1203-1219)
            self.bias = nn.Parameter(torch.zeros(shape[0]))


class _GAttention(nn.Module):
    def __init__(self, H, kv, b):
        super().__init__()
        self.query, self.key, self.value, self.out = _P(H, H, bias=b), _P(H, kv, bias=b), _P(H, kv, bias=b), _P(H, H, bias=b)


class _GFeedForward(nn.Module):
    def __init__(self, H, I, normformer, b, nb):
        super().__init__()
        self.pre_mlp_layer_norm = _P(H, ones=True, bias=b)      # a LayerNorm whatever norm_type says (:768-770): biased under use_bias
        self.wi_0, self.wi_1 = _P(I, H, bias=b), _P(I, H, bias=b)
        if normformer:
            self.mid_mlp_layer_norm = _P(I, ones=True, bias=nb)
        self.wo = _P(H, I, bias=b)


class _GLayer(nn.Module):
    """b: use_bias (every Linear);  nb: use_bias and norm_type == "layernorm" (norm_cls norms; RMSNorm has no bias)"""

    def __init__(self, H, I, kv, cross, normformer, b, nb):
        super().__init__()
        self.attn_layer_norm = _P(H, ones=True, bias=nb)
        self.attention = _GAttention(H, H, b)
        if normformer:
            self.post_attn_layer_norm = _P(H, ones=True, bias=nb)
        self.ffn = _GFeedForward(H, I, normformer, b, nb)
        if cross:
            self.crossattn_layer_norm = _P(H, ones=True, bias=nb)
            self.crossattention = _GAttention(H, kv, b)
            if normformer:
                self.post_crossattn_layer_norm = _P(H, ones=True, bias=nb)


class _GEmbed(nn.Module):
    def __init__(self, V, P, H):
        super().__init__()
        self.word_embeddings, self.position_embeddings = _P(V, H), _P(P, H)


class _GMlm(nn.Module):
    def __init__(self, H, V, layernorm, b, nb):
        super().__init__()
        self.mlm_dense = _P(H, H, bias=b)
        if layernorm:
            self.mlm_ln = _P(H, ones=True, bias=nb)
        self.to_logits = _P(V, H, bias=b)


class _GeneralFn(torch.autograd.Function):
    """one autograd node for the whole network; parameter gradients (state-dict order) and, when asked for, the gradient of the
    text states go back to autograd"""

    @staticmethod
    def forward(ctx, model, input_ids, enc, labels, label_smoothing, cond_dropout_prob, cond_uniforms, need_grad, *params):
        model._drop_step_caches()
        model.__dict__["_act_cache_on"] = bool(need_grad)
        try:
            with model._gemm_mode():
                logits, loss, tape = model._gen_forward(input_ids, enc, labels, label_smoothing, cond_dropout_prob, cond_uniforms, need_grad)
        except BaseException:
            model._drop_step_caches()      # a forward that raised keeps no activation copies / operand images alive
            raise
        finally:
            model.__dict__["_act_cache_on"] = False
        ctx.model, ctx.tape = model, tape
        ctx.enc_grad = bool(need_grad and enc is not None and enc.requires_grad)
        ctx.set_materialize_grads(False)
        if loss is None:
            ctx.mark_non_differentiable(logits)       # (eval-mode forward under grad: plain, non-differentiable logits)
            return logits
        ctx.mark_non_differentiable(logits)
        return logits, loss

    @staticmethod
    def backward(ctx, g_logits, g_loss=None):
        if ctx.tape is None:
            raise MuseHipError("backward called on a forward that ran without grad")
        if g_loss is None:
            raise MuseHipError("MaskGitTransformer (text-conditioned / general form): only the loss is differentiable (pass labels)")
        model = ctx.model
        model.__dict__["_dw_pending"] = []          # (a backward that raised may have left collected products behind)
        model.__dict__["_grads_reported"] = set()
        with model._gemm_mode(backward=True):
            G = model._gen_backward(ctx.tape, g_loss, ctx.enc_grad)
        ctx.tape = None
        model._drop_step_caches()
        grads = tuple(G.get(name) for name, _ in model.named_parameters())
        return (None, None, G.get("__encoder_hidden_states__"), None, None, None, None, None) + grads


class GeneralMaskGitEngine(TapeOps):
    """mixed into muse.MaskGitTransformer; active when `self._general` is set"""

    def _gen_build(self):
        c = self.config
        H, I, V = c.hidden_size, c.intermediate_size, c.vocab_size
        cross = bool(c.add_cross_attention)
        kv = c.encoder_hidden_size
        b = bool(c.use_bias)                              # a bias on every nn.Linear (:170-176, :770-778, :973-977, :1155, :1197) ...
        nb = b and c.norm_type == "layernorm"             # ... and on every LayerNorm (:130); RMSNorm has none
        self.__dict__["_use_bias"] = b
        self.embed = _GEmbed(V, c.max_position_embeddings, H)   # (Embed gets no bias: its norm / projection are never built, :1143-1152)
        if c.add_cross_attention is not None and c.project_encoder_hidden_states:      # (:1143: `is not None`, as written)
            self.encoder_proj = _P(H, c.encoder_hidden_size, bias=b)
            self.encoder_proj_layer_norm = _P(H, ones=True, bias=nb)
            kv = H
        self.transformer_layers = nn.ModuleList([_GLayer(H, I, kv, cross, bool(c.use_normformer), b, nb)
                                                 for _ in range(c.num_hidden_layers)])
        if c.use_encoder_layernorm:
            self.encoder_layer_norm = _P(H, ones=True, bias=nb)
        if c.use_mlm_layer:
            self.mlm_layer = _GMlm(H, self.output_size, bool(c.use_mlm_layernorm), b, nb)
        else:
            self.to_logits = _P(self.output_size, H, bias=b)
        self.compute_dtype = torch.float32
        self._side_stream = None
        for name, p in self.named_parameters():      # reference :1203-1219
            if p.dim() > 1:
                nn.init.trunc_normal_(p.data, std=c.initializer_range)

    # ---- small tape helpers ----------------------------------------------------------------------------------------------
    def _post_norm_add(self, a, mod, x, mode):
        """x + norm(a) * w   (NormFormer's post-attention norms, reference :882-884 / :896-898)"""
        eps = float(self.config.layer_norm_eps)
        if mode == 1:
            y, mean, rstd = ops.layernorm_fwd(a, self._f(mod.weight), eps, torch.float32, residual=x)
            if getattr(mod, "bias", None) is not None:
                ops.add_rowvec_(y, self._f(mod.bias))
            return y, dict(a=a, mean=mean, rstd=rstd)
        n, _ = ops.norm_res_fwd(a, self._f(mod.weight), eps, 0)
        return x + n, dict(a=a)            # (RMSNorm + NormFormer: no shipped config; one ATen add)

    def _post_norm_add_bwd(self, dy, sv, mod, name, G, mode):
        """-> d(a); d(x) = dy"""
        eps = float(self.config.layer_norm_eps)
        if getattr(mod, "bias", None) is not None:
            G[name + ".bias"] = ops.bias_grad(dy)
        if mode == 1:
            dw = torch.empty_like(self._f(mod.weight))
            da = ops.layernorm_bwd(dy, sv["a"], self._f(mod.weight), sv["mean"], sv["rstd"], torch.float32, dw, False)
            G[name + ".weight"] = dw
            return da
        da, dw = ops.norm_res_bwd(dy, sv["a"], self._f(mod.weight), eps, 0)
        G[name + ".weight"] = dw
        return da

    def _attn_block_g(self, x, ctx, Skv, norm_mod, att, post_mod, B, S, nh, mode, drop):
        n, _ = self._norm(x, norm_mod, mode)
        kv = n if ctx is None else ctx
        if post_mod is not None:
            a, sa = self._attention(n, kv, att, B, S, Skv, nh, drop=drop)
            y, sp = self._post_norm_add(a, post_mod, x, mode)
        else:
            y, sa = self._attention(n, kv, att, B, S, Skv, nh, residual=x, drop=drop)
            sp = None
        return y, dict(x=x, sa=sa, sp=sp)

    def _attn_block_g_bwd(self, dy, sv, norm_mod, att, post_mod, name, att_name, post_name, G, mode, self_attn):
        """-> (d(x), d(ctx) or None)"""
        da = self._post_norm_add_bwd(dy, sv["sp"], post_mod, post_name, G, mode) if post_mod is not None else dy
        dn, dctx = self._attention_bwd(da, sv["sa"], att, att_name, G, self_attn=self_attn)
        dx = self._norm_bwd(dn, sv["x"], norm_mod, name, G, mode=mode, dpre=dy)      # + the residual branch
        return dx, dctx

    # ---- forward -------------------------------------------------------------------------------------------------------------
    def _gen_forward(self, input_ids, enc, labels, label_smoothing, cond_dropout_prob, cond_uniforms, need_grad):
        c = self.config
        B, S = input_ids.shape
        H, nh, V = c.hidden_size, c.num_attention_heads, self.output_size
        mode = 1 if c.norm_type == "layernorm" else 0
        nf = bool(c.use_normformer)
        f = self._f
        dev = input_ids.device
        T = {}
        ids = input_ids.contiguous()
        x = ops.embed_fwd(ids, f(self.embed.word_embeddings.weight), f(self.embed.position_embeddings.weight))
        pd_h = float(c.hidden_dropout) if self.training else 0.0
        pd_a = float(c.attention_dropout) if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,))) if (pd_h > 0.0 or pd_a > 0.0) else 0
        site = lambda li, k: ((li * 4 + k + 1) << 40)   # noqa: E731  (k = 0 self-attention, 1 cross-attention, 2 feed-forward; 0 = embeddings)
        if pd_h > 0.0:
            ops.dropout(x, pd_h, seed, 0, out=x)
        # text states (:1239-1247)
        L = 0
        e_in = e0 = e1 = e2 = u = None
        keep = 1.0
        if enc is not None:
            L = enc.shape[1]
            e_in = enc.reshape(B * L, -1).float().contiguous()
            if c.project_encoder_hidden_states:
                e0 = self._lin(e_in, self.encoder_proj)
                e1, _ = self._norm(e0, self.encoder_proj_layer_norm, mode)
            else:
                e1 = e_in
            e2 = e1
            if self.training and cond_dropout_prob > 0.0:
                keep = 1.0 - float(cond_dropout_prob)     # prob_mask_like(.., 1 - p): the states of an image survive iff u < 1 - p
                u = cond_uniforms if cond_uniforms is not None else torch.rand(B, device=dev)
                e2 = ops.cond_dropout(e1.view(B, L, -1), torch.zeros(e1.numel() // B, device=dev), u.to(dev), keep).view(B * L, -1)
        elif c.add_cross_attention:
            raise ValueError("If `add_cross_attention` is True, `encoder_hidden_states` should be provided.")
        T["layers"] = []
        for li, lyr in enumerate(self.transformer_layers):
            da = (pd_a, seed, site(li, 0)) if pd_a > 0.0 else None
            x1, s1 = self._attn_block_g(x, None, S, lyr.attn_layer_norm, lyr.attention, lyr.post_attn_layer_norm if nf else None,
                                        B, S, nh, mode, da)
            s2 = None
            if e2 is not None and hasattr(lyr, "crossattention"):
                dc = (pd_a, seed, site(li, 1)) if pd_a > 0.0 else None
                x1, s2 = self._attn_block_g(x1, e2, L, lyr.crossattn_layer_norm, lyr.crossattention,
                                            lyr.post_crossattn_layer_norm if nf else None, B, S, nh, mode, dc)
            n3, _ = self._norm(x1, lyr.ffn.pre_mlp_layer_norm, 1)                    # always a LayerNorm (:768-770)
            w01 = self._w2(lyr.ffn.wi_0, lyr.ffn.wi_1)
            bf_chain = self.compute_dtype == torch.bfloat16 and not nf and pd_h == 0.0
            b01 = self._b(lyr.ffn.wi_0, lyr.ffn.wi_1)
            ab = ops.linear(self._c(n3), w01, out_dtype=torch.bfloat16, bias=b01) if bf_chain else self._mm(n3, w01, bias=b01)
            g = ops.glu_fwd(ab)                                                       # gelu(wi_0 x) * (wi_1 x)  (:789-792)
            gm = g
            if nf:
                gm, _ = self._norm(g, lyr.ffn.mid_mlp_layer_norm, mode)
            if pd_h > 0.0:
                gm = ops.dropout(gm, pd_h, seed, site(li, 2))
            x2 = self._lin(gm, lyr.ffn.wo, residual=x1)
            T["layers"].append(dict(s1=s1, s2=s2, x1=x1, n3=n3, w01=w01, ab=ab, g=g, gm=gm))
            x = x2
        xe = x
        if c.use_encoder_layernorm:
            xe, _ = self._norm(x, self.encoder_layer_norm, mode)
        d = gd = None
        if c.use_mlm_layer:
            d = self._lin(xe, self.mlm_layer.mlm_dense)
            gd = ops.gelu_fwd(d)
            gl = gd
            if c.use_mlm_layernorm:
                gl, _ = self._norm(gd, self.mlm_layer.mlm_ln, mode)
            head = self.mlm_layer.to_logits
        else:
            gl, head = xe, self.to_logits
        Vp = (V + 7) // 8 * 8
        w2 = self._w2(head)
        logits_p = torch.empty((B * S, Vp), dtype=torch.float32, device=dev)
        gl_op, w2 = self._pair(gl, w2)
        ops.gemm(gl_op, w2, logits_p, B * S, V, H, lda=H, ldb=H, ldc=Vp, bias=self._b(head))
        logits = logits_p.view(B, S, Vp) if Vp == V else logits_p[:, :V].contiguous().view(B, S, V)
        loss = None
        if labels is not None:
            lab = labels.reshape(-1).contiguous()
            self.__dict__["_loss_rows"] = lab.numel()        # ("f16" mode: bounds d(logits), tape_ops.f16_grad_scale_for)
            loss_out, lse = ops.cross_entropy_fwd(logits_p, lab, float(label_smoothing), vocab=V)
            loss = loss_out[0]
            T["ce"] = dict(lab=lab, lse=lse, loss_out=loss_out, ls=float(label_smoothing))
        if not need_grad:
            return logits, loss, None
        T.update(B=B, S=S, L=L, ids=ids, e_in=e_in, e0=e0, e1=e1, u=u, keep=keep, x_last=x, xe=xe, d=d, gd=gd, gl=gl, head=head,
                 logits_p=logits_p, V=V, Vp=Vp, drop=(seed, pd_h, pd_a), mode=mode, nf=nf, site=site)
        return logits, loss, T

    # ---- backward ------------------------------------------------------------------------------------------------------------
    def _gen_backward(self, T, g_loss, want_enc_grad):
        c = self.config
        B, S, L, V, Vp, mode, nf = T["B"], T["S"], T["L"], T["V"], T["Vp"], T["mode"], T["nf"]
        seed, pd_h, pd_a = T["drop"]
        site = T["site"]
        G = {}
        ce = T["ce"]
        dev = T["logits_p"].device
        go = g_loss.reshape(1).to(torch.float32).contiguous()
        dl = ops.cross_entropy_bwd(T["logits_p"], ce["lab"], ce["lse"], ce["loss_out"], go, ce["ls"], torch.float32, vocab=V)
        head = T["head"]
        hname = "mlm_layer.to_logits" if c.use_mlm_layer else "to_logits"
        w2 = self._w2(head)
        dlc = self._c(dl)
        G[hname + ".weight"] = self._mm_dw(dlc, T["gl"], w2.shape, M=V, lda=Vp).view(head.weight.shape)
        if self._use_bias:
            G[hname + ".bias"] = ops.bias_grad(dl, cols=V)
        dgl = self._mm_dx(dlc, w2, lda=Vp)
        if c.use_mlm_layer:
            dgd = self._norm_bwd(dgl, T["gd"], self.mlm_layer.mlm_ln, "mlm_layer.mlm_ln", G, mode=mode) if c.use_mlm_layernorm else dgl
            dd = ops.gelu_bwd(T["d"], dgd)
            dxe = self._lin_bwd(dd, T["xe"], self.mlm_layer.mlm_dense, "mlm_layer.mlm_dense", G)
        else:
            dxe = dgl
        dx = self._norm_bwd(dxe, T["x_last"], self.encoder_layer_norm, "encoder_layer_norm", G, mode=mode) if c.use_encoder_layernorm else dxe
        denc = None
        for li in reversed(range(c.num_hidden_layers)):
            lyr, sv = self.transformer_layers[li], T["layers"][li]
            nm = f"transformer_layers.{li}"
            # feed-forward: x2 = x1 + wo gm
            if sv["ab"].dtype == torch.bfloat16:      # bf16 GLU chain (no mid norm, no dropout): mirrors the forward
                wo2, dxb = self._w2(lyr.ffn.wo), self._c(dx)
                G[nm + ".ffn.wo.weight"] = self._mm_dw(dxb, sv["gm"], wo2.shape).view(lyr.ffn.wo.weight.shape)
                if self._use_bias:
                    G[nm + ".ffn.wo.bias"] = ops.bias_grad(dx)
                dg = ops.linear_dgrad(dxb, wo2)
            else:
                dgm = self._lin_bwd(dx, sv["gm"], lyr.ffn.wo, nm + ".ffn.wo", G)
                if pd_h > 0.0:
                    ops.dropout(dgm, pd_h, seed, site(li, 2), out=dgm)
                dg = self._norm_bwd(dgm, sv["g"], lyr.ffn.mid_mlp_layer_norm, nm + ".ffn.mid_mlp_layer_norm", G, mode=mode) if nf else dgm
            dab = ops.glu_bwd(sv["ab"], dg)
            gw01 = self._mm_dw(dab, sv["n3"], sv["w01"].shape)
            I = gw01.shape[0] // 2
            G[nm + ".ffn.wi_0.weight"], G[nm + ".ffn.wi_1.weight"] = gw01[:I], gw01[I:]
            if self._use_bias:
                gb01 = ops.bias_grad(dab)
                G[nm + ".ffn.wi_0.bias"], G[nm + ".ffn.wi_1.bias"] = gb01[:I], gb01[I:]
            dn3 = self._mm_dx(dab, sv["w01"])
            dx1 = self._norm_bwd(dn3, sv["x1"], lyr.ffn.pre_mlp_layer_norm, nm + ".ffn.pre_mlp_layer_norm", G, mode=1, dpre=dx)
            if sv["s2"] is not None:
                dx1, dctx = self._attn_block_g_bwd(dx1, sv["s2"], lyr.crossattn_layer_norm, lyr.crossattention,
                                                   lyr.post_crossattn_layer_norm if nf else None, nm + ".crossattn_layer_norm",
                                                   nm + ".crossattention", nm + ".post_crossattn_layer_norm", G, mode, False)
                denc = dctx if denc is None else denc.add_(dctx)
            dx, _ = self._attn_block_g_bwd(dx1, sv["s1"], lyr.attn_layer_norm, lyr.attention, lyr.post_attn_layer_norm if nf else None,
                                           nm + ".attn_layer_norm", nm + ".attention", nm + ".post_attn_layer_norm", G, mode, True)
            T["layers"][li] = None
            self._report_grads(G)        # (data-parallel: this layer's gradients go to the reducer's buckets while backward continues)
        if pd_h > 0.0:
            ops.dropout(dx, pd_h, seed, 0, out=dx)
        gw = torch.zeros_like(self._f(self.embed.word_embeddings.weight))
        gp = torch.zeros_like(self._f(self.embed.position_embeddings.weight))
        ops.embed_bwd(T["ids"], dx, gw, gp, False)
        G["embed.word_embeddings.weight"], G["embed.position_embeddings.weight"] = gw, gp
        # text states: gradients of the projection, and of the input when it asks for one
        if denc is not None and (c.project_encoder_hidden_states or want_enc_grad):
            if T["u"] is not None:
                denc = ops.cond_dropout(denc.view(B, L, -1), torch.zeros(denc.numel() // B, device=dev), T["u"].to(dev), T["keep"]).view(B * L, -1)
            if c.project_encoder_hidden_states:
                de0 = self._norm_bwd(denc, T["e0"], self.encoder_proj_layer_norm, "encoder_proj_layer_norm", G, mode=mode)
                de_in = self._lin_bwd(de0, T["e_in"], self.encoder_proj, "encoder_proj", G, need_dx=want_enc_grad)
            else:
                de_in = denc
            if want_enc_grad:
                G["__encoder_hidden_states__"] = de_in.view(B, L, -1)
        self._report_grads(G, final=True)
        if self.__dict__.pop("_side_busy", False):
            torch.cuda.current_stream(dev).wait_stream(self._side_stream)   # every weight gradient is complete before autograd sees it
        return G

    def _gen_call(self, input_ids, encoder_hidden_states, labels, label_smoothing, cond_dropout_prob, cond_dropout_uniforms):
        if not input_ids.is_cuda or (encoder_hidden_states is not None and not encoder_hidden_states.is_cuda):
            raise MuseHipError("MaskGitTransformer (MI355X build) has no CPU path: move the model and inputs to the GPU")
        params = [p for _, p in self.named_parameters()]
        need_grad = torch.is_grad_enabled() and labels is not None and any(p.requires_grad for p in params)
        if labels is None and torch.is_grad_enabled() and self.training and any(p.requires_grad for p in params):
            # this engine differentiates its internal loss only (the hand-written backward starts at the cross entropy); a
            # logits-only forward in training mode under grad would hand back logits whose backward cannot run - say so here
            raise MuseHipError("MaskGitTransformer (text-conditioned / general form): only the internal loss is differentiable - pass "
                               "`labels` (and `label_smoothing`), or run the logits-only forward under torch.no_grad() / model.eval()")
        return _GeneralFn.apply(self, input_ids, encoder_hidden_states, labels, float(label_smoothing), float(cond_dropout_prob),
                                cond_dropout_uniforms, need_grad, *params)
