"""SAM 2.1 image path on the cvmi355 kernels: weight preparation (LoRA merge, constant folding),
launch plan, and the reference's boundary objects.

What it replaces:
  * `get_modified_sam2(...)` -> model with `.load_state_dict`, `.eval`, `.sam2_model.image_size`,
    `__call__(x[B,3,1024,1024]) -> (high_res[B,1,1024,1024], low_res[B,1,256,256], iou[B,1])`
    (/root/reference/src/sam2_infer.py:277-410, :220-275; caller circuit_analyzer.py:203-242, :349-351)
  * `SAM2Transforms(resolution, mask_threshold, max_hole_area, max_sprinkle_area)` with `__call__`
    and `postprocess_masks` (sam2_infer.py:29-128; caller circuit_analyzer.py:245-250, :347, :354)
  * `infer_masks(images)`: the batched entry point named by BASELINE.json's north_star.

Weight preparation (host, fp32, once): LoRA adapters merged (W' = W + (alpha/r) B A, rows B12); the
FPN lateral convs composed with conv_s0 / conv_s1 (no 256-channel 256^2 map is ever materialised);
FPN level 2 = lateral(stage 3) + nearest-2x lateral(stage 4) as ONE two-source GEMM with the learned
dense prompt added in its epilogue; every "+ positional encoding" of the two-way transformer folded
into constant post-projection residuals; Hiera's bicubic position embedding precomputed.

Numerics: the residual stream, LayerNorm statistics, softmax and all accumulators are fp32; in F16
mode GEMM / attention operands are fp16.  F32 mode runs everything on exact-f32 MFMA (parity mode).
"""
import hashlib
import math
import os
import threading

import numpy as np
import torch
import torch.nn.functional as TF

from . import _lib
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, BF16, F16, F32
from .engine import (ESIZE, TORCH_DTYPE, is16, Buf, PackedConv, PackedHieraMlp, PackedTokLinear, Plan, Rows, hiera_mlp_supported, make_attn_desc, op_attention,
                     op_call, op_cast, op_conv, op_hiera_mlp, op_layernorm, op_maxpool2, op_tok_linear, op_tok_linear_pool, require_gpu, row_stats_supported,
                     tok_linear_stats_parts, tok_linear_supported)

LOG2E = 1.4426950408889634
HIERA_L = dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8))
HIERA_T = dict(embed_dim=96, num_heads=1, stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9), window_spec=(8, 4, 14, 7))

# circuit_analyzer.py:156-199 -- the 36 LoRA-wrapped modules of the fine-tuned checkpoint
LORA_TARGETS_REFERENCE = (
    [f"sam_mask_decoder.transformer.layers.{l}.{a}.{p}" for l in (0, 1) for a in ("self_attn", "cross_attn_token_to_image")
     for p in ("k_proj", "q_proj", "v_proj", "out_proj")]
    + [f"sam_mask_decoder.transformer.layers.{l}.mlp.layers.{j}" for l in (0, 1) for j in (0, 1)]
    + ["sam_mask_decoder.iou_prediction_head.layers.2", "sam_mask_decoder.conv_s0", "sam_mask_decoder.conv_s1",
       "image_encoder.neck.convs.2.conv", "image_encoder.neck.convs.3.conv",
       "image_encoder.trunk.blocks.44.attn.qkv", "image_encoder.trunk.blocks.44.mlp.layers.0", "image_encoder.trunk.blocks.44.proj",
       "image_encoder.trunk.blocks.47.attn.qkv", "image_encoder.trunk.blocks.47.mlp.layers.0"]
    + [f"sam_mask_decoder.transformer.layers.{l}.cross_attn_image_to_token.{p}" for l in (0, 1) for p in ("q_proj", "k_proj", "v_proj")])


# ---- parameter sources ---------------------------------------------------------------------------------
class SamSyntheticParams:
    """Seeded synthetic weights in the fine-tuned checkpoint's own key format (PEFT names for LoRA
    targets, wrapper parameters at the top level).  `state_dict()` loads strictly into the oracle."""

    def __init__(self, seed=0, lora_targets=LORA_TARGETS_REFERENCE, r=4, alpha=16, std=0.02):
        self.seed, self.targets, self.r, self.scaling, self.std, self.sd = seed, set(lora_targets), r, alpha / r, std, {}

    def _t(self, name, shape, kind="w"):
        if name in self.sd:
            assert tuple(self.sd[name].shape) == tuple(shape), (name, self.sd[name].shape, shape)
            return self.sd[name]
        h = int.from_bytes(hashlib.sha256(f"{self.seed}:{name}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF
        g = torch.Generator().manual_seed(h)
        if kind == "gamma":
            t = torch.empty(shape).uniform_(0.8, 1.2, generator=g)
        elif kind == "unit":
            t = torch.empty(shape).normal_(0, 1.0, generator=g)
        elif kind == "refine":
            t = torch.empty(shape).normal_(0, 0.2, generator=g)
        else:
            t = torch.empty(shape).normal_(0, self.std, generator=g).clamp_(-2 * self.std, 2 * self.std)
        self.sd[name] = t
        return t

    def weight(self, mod, shape):
        """Merged weight of module `mod` (Linear [out,in] or Conv [out,in,kh,kw])."""
        if mod in self.targets:
            w = self._t(f"{mod}.base_layer.weight", shape)
            a_shape = (self.r, shape[1]) + tuple(shape[2:])
            b_shape = (shape[0], self.r) + (1,) * (len(shape) - 2)
            A = self._t(f"{mod}.lora_A.default.weight", a_shape)
            Bm = self._t(f"{mod}.lora_B.default.weight", b_shape)
            delta = (Bm.reshape(shape[0], self.r) @ A.reshape(self.r, -1)).reshape(shape)
            return w + self.scaling * delta
        return self._t(f"{mod}.weight", shape)

    def bias(self, mod, n):
        return self._t(f"{mod}.base_layer.bias" if mod in self.targets else f"{mod}.bias", (n,))

    def tensor(self, name, shape, kind="w"):
        return self._t(name, shape, kind)

    def state_dict(self):
        return dict(self.sd)


class SamBlankParams:
    """A rank that holds NO checkpoint: every tensor reads as zeros, so `Sam2Weights` allocates the packed buffers at their final shapes
    (no random generation, no file) and `distributed.broadcast_weights` fills them from the rank that read the checkpoint."""

    def weight(self, mod, shape):
        return torch.zeros(shape)

    def bias(self, mod, n):
        return torch.zeros(n)

    def tensor(self, name, shape, kind="w"):
        return torch.zeros(shape)

    def state_dict(self):
        return {}


class SamStateDictParams:
    """A real checkpoint: flat state_dict (optionally under 'state_dict'), PEFT key names with the
    `sam2_model.base_model.model.` prefix (sam2_infer.py:396), wrapper parameters at the top level."""

    PREFIXES = ("sam2_model.base_model.model.", "sam2_model.")

    def __init__(self, sd, r=4, alpha=16):
        self.scaling = alpha / r
        self.sd = {}
        for k, v in sd.items():
            for p in self.PREFIXES:
                if k.startswith(p):
                    k = k[len(p):]
                    break
            self.sd[k] = v.detach().float().cpu()

    def _get(self, name, shape):
        t = self.sd[name]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != expected {tuple(shape)}")
        return t

    def weight(self, mod, shape):
        if f"{mod}.base_layer.weight" in self.sd:
            w = self._get(f"{mod}.base_layer.weight", shape)
            A, Bm = self.sd[f"{mod}.lora_A.default.weight"], self.sd[f"{mod}.lora_B.default.weight"]
            r = A.shape[0]
            return w + self.scaling * (Bm.reshape(shape[0], r) @ A.reshape(r, -1)).reshape(shape)
        return self._get(f"{mod}.weight", shape)

    def bias(self, mod, n):
        k = f"{mod}.base_layer.bias" if f"{mod}.base_layer.bias" in self.sd else f"{mod}.bias"
        return self._get(k, (n,))

    def tensor(self, name, shape, kind="w"):
        return self._get(name, shape)


class SamBaseCheckpointParams(SamStateDictParams):
    """The SAM 2 base checkpoint (`{'model': state_dict}`, plain upstream keys) that `build_sam2` loads
    (sam2_infer.py:333).  It has no LoRA tensors and none of the wrapper's own parameters; those start, as in the
    reference before the fine-tuned state dict is loaded, from `torch.randn` (sam2_infer.py:207-209) / default conv
    init -- seeded here so that runs are reproducible."""

    WRAPPER = ("dense_embedding", "sparse_embedding", "refinement_layer.")

    def __init__(self, sd, seed=0):
        super().__init__(sd.get("model", sd) if isinstance(sd, dict) else sd)
        self.seed = seed

    def tensor(self, name, shape, kind="w"):
        if name in self.sd:
            return self._get(name, shape)
        if name.startswith(self.WRAPPER):
            h = int.from_bytes(hashlib.sha256(f"{self.seed}:{name}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF
            g = torch.Generator().manual_seed(h)
            t = torch.empty(shape).normal_(0, 1.0 if "embedding" in name else 0.1, generator=g)
            self.sd[name] = t
            return t
        raise KeyError(name)


# ---- prepared weights -----------------------------------------------------------------------------------------
def _lin(w):
    return w.reshape(w.shape[0], w.shape[1], 1, 1)


class Sam2Weights:
    def __init__(self, params, hiera=HIERA_L, image_size=1024, dtype=F16, device="cuda", use_refinement=True,
                 refinement_kernels=(3, 5, 7, 11), embedding_r=4):
        self.p, self.hiera, self.image_size, self.dtype, self.device = params, hiera, image_size, dtype, device
        self.use_refinement, self.kernels = use_refinement, tuple(refinement_kernels)
        self.pc, self.ln, self.const, self.mlp, self.tl = {}, {}, {}, {}, {}
        self.fused_mlp = os.environ.get("CVMI_SAM_FUSED_MLP", "1") != "0"      # (the switches are for A/B measurements)
        self.use_tok = os.environ.get("CVMI_SAM_TOKLIN", "1") != "0"
        # Hiera q rows carry scale * log2(e) (16-bit plans): see _linear(row_scale=) and cvmi_attn_desc.q_log2
        self.q_log2 = is16(dtype) and os.environ.get("CVMI_SAM_QLOG2", "1") != "0"
        self.param_bytes = 0
        self.flops_per_image = 0
        self._trunk()
        self._neck_and_prompts(embedding_r)
        self._decoder()
        self._refinement()

    # -- helpers
    def _pack(self, key, w4, b):
        pc = PackedConv(w4, b, self.dtype, self.device)
        self.pc[key] = pc
        self.param_bytes += pc.param_bytes
        return pc

    def _linear(self, key, mod, cout, cin, tok=False, row_scale=None):
        w, b = self.p.weight(mod, (cout, cin)), self.p.bias(mod, cout)
        if row_scale is not None:
            # (rows, factor): the attention scale * log2(e) folded into the q rows of a qkv projection, in fp32 before the weights are rounded
            # to the operand type -- the attention kernels then take exp2 of q k^T as it stands (cvmi_attn_desc.q_log2)
            n, f = row_scale
            w, b = w.clone(), b.clone()
            w[:n] *= f
            b[:n] *= f
        if tok and self.use_tok and is16(self.dtype) and tok_linear_supported(cin, self.dtype, 256):
            # short-K Hiera linears also in the token-stationary kernel's fragment order (tok_linear.hip); the plan picks it
            # whenever its row count is a multiple of 256
            self.tl[key] = PackedTokLinear(w, b, self.device, self.dtype)
        return self._pack(key, _lin(w), b)

    def _norm(self, key, mod, c):
        self.ln[key] = (self.p.tensor(f"{mod}.weight", (c,), "gamma").float().to(self.device),
                        self.p.tensor(f"{mod}.bias", (c,)).float().to(self.device))

    def _trunk(self):
        h = self.hiera
        E, stages, ws = h["embed_dim"], h["stages"], h["window_spec"]
        T = "image_encoder.trunk"
        # PatchEmbed 7x7 / s4 / pad 3 on the space-to-depth(4) image: window row ky = 0..2 sits in block row -1 (sub-row
        # ky + 1), ky = 3..6 in block row 0 (sub-row ky - 3); same for columns -> a 2x2 conv over 48 channels
        w7 = self.p.weight(f"{T}.patch_embed.proj", (E, 3, 7, 7))
        w2 = torch.zeros(E, 48, 2, 2)
        m = {k: ((0, k + 1) if k < 3 else (1, k - 3)) for k in range(7)}
        for ky in range(7):
            for kx in range(7):
                (ty, sy), (tx, sx) = m[ky], m[kx]
                w2[:, (sy * 4 + sx) * 3:(sy * 4 + sx) * 3 + 3, ty, tx] = w7[:, :, ky, kx]
        self._pack("patch_embed", w2, self.p.bias(f"{T}.patch_embed.proj", E))
        g = self.image_size // 4
        pos = TF.interpolate(self.p.tensor(f"{T}.pos_embed", (1, E, 7, 7)), size=(g, g), mode="bicubic")
        win = self.p.tensor(f"{T}.pos_embed_window", (1, E, ws[0], ws[0]))
        pos = pos + win.tile([1, 1, g // ws[0], g // ws[0]])
        self.const["pos_embed"] = pos.permute(0, 2, 3, 1).reshape(g * g, E).contiguous().float().to(self.device)   # f32 residual stream
        stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
        q_pool_blocks = [x + 1 for x in stage_ends[:-1]][:3]
        self.blocks, self.stage_ends = [], stage_ends
        cur, dim, heads = 1, E, h["num_heads"]
        for i in range(sum(stages)):
            dim_out, window = dim, ws[cur - 1]
            if i in h["global_att_blocks"]:
                window = 0
            if i - 1 in stage_ends:
                dim_out, heads, cur = dim * 2, heads * 2, cur + 1
            b = f"{T}.blocks.{i}"
            self._norm(f"b{i}.norm1", f"{b}.norm1", dim)
            self._linear(f"b{i}.qkv", f"{b}.attn.qkv", 3 * dim_out, dim, tok=True,
                         row_scale=(dim_out, (dim_out // heads) ** -0.5 * LOG2E) if self.q_log2 else None)
            self._linear(f"b{i}.proj", f"{b}.attn.proj", dim_out, dim_out, tok=True)
            self._norm(f"b{i}.norm2", f"{b}.norm2", dim_out)
            if self.fused_mlp and hiera_mlp_supported(dim_out, self.dtype):
                # stages 1 / 2: norm2 + fc1 + GELU + fc2 + residual as ONE launch (hiera_mlp.hip); weights in fragment order
                self.mlp[f"b{i}"] = PackedHieraMlp(self.p.weight(f"{b}.mlp.layers.0", (4 * dim_out, dim_out)), self.p.bias(f"{b}.mlp.layers.0", 4 * dim_out),
                                                   self.p.weight(f"{b}.mlp.layers.1", (dim_out, 4 * dim_out)), self.p.bias(f"{b}.mlp.layers.1", dim_out),
                                                   self.device, self.dtype)
                self.param_bytes += self.mlp[f"b{i}"].param_bytes
            else:
                self._linear(f"b{i}.fc1", f"{b}.mlp.layers.0", 4 * dim_out, dim_out, tok=True)
                self._linear(f"b{i}.fc2", f"{b}.mlp.layers.1", dim_out, 4 * dim_out)
            if dim != dim_out:
                self._linear(f"b{i}.dimproj", f"{b}.proj", dim_out, dim, tok=True)
            self.blocks.append(dict(dim=dim, dim_out=dim_out, heads=heads, window=window, q_pool=i in q_pool_blocks))
            if window > 0 and i in q_pool_blocks and window % 2:
                raise ValueError("q-pool block with odd window")
            dim = dim_out
        self.stage_dims = [E * 2 ** s for s in range(len(stages))]

    def _neck_and_prompts(self, embedding_r):
        N, D = "image_encoder.neck", "sam_mask_decoder"
        c = self.stage_dims                                   # [144, 288, 576, 1152]
        wn = [self.p.weight(f"{N}.convs.{j}.conv", (256, c[3 - j], 1, 1)).reshape(256, -1) for j in range(4)]
        bn = [self.p.bias(f"{N}.convs.{j}.conv", 256) for j in range(4)]
        ws0, bs0 = self.p.weight(f"{D}.conv_s0", (32, 256, 1, 1)).reshape(32, 256), self.p.bias(f"{D}.conv_s0", 32)
        ws1, bs1 = self.p.weight(f"{D}.conv_s1", (64, 256, 1, 1)).reshape(64, 256), self.p.bias(f"{D}.conv_s1", 64)
        # level 0 (256^2, 144 ch): conv_s0 o lateral ; level 1 (128^2, 288 ch): conv_s1 o lateral
        self._pack("feat_s0", _lin(ws0 @ wn[3]), ws0 @ bn[3] + bs0)
        self._pack("feat_s1", _lin(ws1 @ wn[2]), ws1 @ bn[2] + bs1)
        if self.use_tok and is16(self.dtype):                    # ... and for tok_linear's plain-f32-input form: the stage outputs need no cast pass
            for key, w_, b_ in (("feat_s0", ws0 @ wn[3], ws0 @ bn[3] + bs0), ("feat_s1", ws1 @ wn[2], ws1 @ bn[2] + bs1)):
                if tok_linear_supported(w_.shape[1], self.dtype, 256) and w_.shape[1] <= 288:
                    self.tl[key] = PackedTokLinear(w_, b_, self.device, self.dtype)
        # level 2 (64^2): lateral(stage 3, 576) + nearest2x(lateral(stage 4, 1152)) as one K-concatenated GEMM
        self._pack("embed", _lin(torch.cat((wn[1], wn[0]), 1)), bn[1] + bn[0])
        fs = self.image_size // 16
        e1 = self.p.tensor("dense_embedding1", (1, 256, embedding_r), "unit")
        e2 = self.p.tensor("dense_embedding2", (1, embedding_r, fs * fs), "unit")
        dense = (e1 @ e2).view(256, fs * fs).t().contiguous()                       # [pixels, 256]  (sam2_infer.py:250)
        self.const["dense"] = dense.float().to(self.device)
        G = self.p.tensor("sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", (2, 128), "unit")
        yy, xx = torch.meshgrid((torch.arange(fs, dtype=torch.float32) + 0.5) / fs, (torch.arange(fs, dtype=torch.float32) + 0.5) / fs, indexing="ij")
        cpe = 2 * math.pi * ((2 * torch.stack((xx, yy), -1) - 1) @ G)
        self.key_pe = torch.cat((torch.sin(cpe), torch.cos(cpe)), -1).reshape(fs * fs, 256)      # get_dense_pe(), [pixels, 256]
        self.sparse = self.p.tensor("sparse_embedding", (1, 32, 256), "unit")[0]
        # upstream prompt encoder (box / point prompts, `infer_masks(images, boxes)`): label table, no-mask dense
        # embedding and the predictor's no_mem_embed, the last two folded into the embed GEMM's bias
        PE = "sam_prompt_encoder"
        try:
            table = torch.cat([self.p.tensor(f"{PE}.not_a_point_embed.weight", (1, 256), "unit")]
                              + [self.p.tensor(f"{PE}.point_embeddings.{i}.weight", (1, 256), "unit") for i in range(4)], 0)
            no_mask = self.p.tensor(f"{PE}.no_mask_embed.weight", (1, 256), "unit")[0]
            no_mem = self.p.tensor("no_mem_embed", (1, 1, 256), "unit").reshape(256)
        except KeyError:
            self.prompt_ok = False                  # a checkpoint stripped of the prompt encoder: learned-prompt path only
            return
        self.prompt_ok = True
        self._pack("embed_box", _lin(torch.cat((wn[1], wn[0]), 1)), bn[1] + bn[0] + no_mask + no_mem)
        self.const["prompt_table"] = table.float().contiguous().to(self.device)
        self.const["gauss"] = G.float().contiguous().to(self.device)

    def _attn(self, key, mod, inner, q_pe=None, k_pe=None):
        """One Attention module.  q / k / v projections packed separately; constants (x + pe) W^T folded."""
        d = {}
        for nm in ("q_proj", "k_proj", "v_proj"):
            d[nm] = (self.p.weight(f"{mod}.{nm}", (inner, 256)), self.p.bias(f"{mod}.{nm}", inner))
        self._linear(f"{key}.out", f"{mod}.out_proj", 256, inner)
        return d

    def _decoder(self):
        D = "sam_mask_decoder"
        dev, od = self.device, TORCH_DTYPE[self.dtype]
        tokens = torch.cat((self.p.tensor(f"{D}.obj_score_token.weight", (1, 256), "unit"), self.p.tensor(f"{D}.iou_token.weight", (1, 256), "unit"),
                            self.p.tensor(f"{D}.mask_tokens.weight", (4, 256), "unit"), self.sparse), 0)          # [38, 256]
        self.tokens = tokens
        self.const["tokens"] = tokens.float().to(dev)
        kpe = self.key_pe
        self.n_tok = tokens.shape[0]
        self._sa_w, self._x_w = {}, {}

        def cres(t):                       # constant residual in the GEMM's output dtype
            return t.contiguous().to(od).to(dev)

        for l in (0, 1):
            L = f"{D}.transformer.layers.{l}"
            # self attention: q, k (+ token pe unless layer 0), v from the same queries -> one GEMM of N = 768
            sa = self._attn(f"l{l}.sa", f"{L}.self_attn", 256)
            self._sa_w[l] = (sa["q_proj"][0], sa["k_proj"][0])
            w = torch.cat([sa[n][0] for n in ("q_proj", "k_proj", "v_proj")], 0)
            b = torch.cat([sa[n][1] for n in ("q_proj", "k_proj", "v_proj")], 0)
            self._pack(f"l{l}.sa.qkv", _lin(w), b)
            if l > 0:
                pe = torch.cat((tokens @ sa["q_proj"][0].t(), tokens @ sa["k_proj"][0].t(), torch.zeros(self.n_tok, 256)), 1)
                self.const[f"l{l}.sa.qkv_pe"] = cres(pe)
            self._norm(f"l{l}.norm1", f"{L}.norm1", 256)
            # token -> image cross attention (internal dim 128)
            t2i = self._attn(f"l{l}.t2i", f"{L}.cross_attn_token_to_image", 128)
            self._pack(f"l{l}.t2i.q", _lin(t2i["q_proj"][0]), t2i["q_proj"][1])
            self.const[f"l{l}.t2i.q_pe"] = cres(tokens @ t2i["q_proj"][0].t())
            self._pack(f"l{l}.t2i.kv", _lin(torch.cat((t2i["k_proj"][0], t2i["v_proj"][0]), 0)), torch.cat((t2i["k_proj"][1], t2i["v_proj"][1]), 0))
            self.const[f"l{l}.t2i.kv_pe"] = cres(torch.cat((kpe @ t2i["k_proj"][0].t(), torch.zeros(kpe.shape[0], 128)), 1))
            self._norm(f"l{l}.norm2", f"{L}.norm2", 256)
            self._linear(f"l{l}.mlp1", f"{L}.mlp.layers.0", 2048, 256)
            self._linear(f"l{l}.mlp2", f"{L}.mlp.layers.1", 256, 2048)
            self._norm(f"l{l}.norm3", f"{L}.norm3", 256)
            # image -> token cross attention: q from keys (+ key pe), k from queries (+ token pe), v from queries
            i2t = self._attn(f"l{l}.i2t", f"{L}.cross_attn_image_to_token", 128)
            self._pack(f"l{l}.i2t.q", _lin(i2t["q_proj"][0]), i2t["q_proj"][1])
            self.const[f"l{l}.i2t.q_pe"] = cres(kpe @ i2t["q_proj"][0].t())
            self._pack(f"l{l}.i2t.kv", _lin(torch.cat((i2t["k_proj"][0], i2t["v_proj"][0]), 0)), torch.cat((i2t["k_proj"][1], i2t["v_proj"][1]), 0))
            self.const[f"l{l}.i2t.kv_pe"] = cres(torch.cat((tokens @ i2t["k_proj"][0].t(), torch.zeros(self.n_tok, 128)), 1))
            self._norm(f"l{l}.norm4", f"{L}.norm4", 256)
            self._x_w[l] = {"t2i.q": t2i["q_proj"][0], "i2t.k": i2t["k_proj"][0]}
        F_ = f"{D}.transformer.final_attn_token_to_image"
        fa = self._attn("final", F_, 128)
        # prompted mode: the token PE is per prompt, so (tokens0 @ W^T) is one run-time GEMM whose column groups are
        # the residuals the constant-folded path reads from `const[...]`
        zeros = lambda n: torch.zeros(n, 256)
        groups = [("l1.sa.qkv_pe", torch.cat((self._sa_w[1][0], self._sa_w[1][1], zeros(256)), 0))]
        for l in (0, 1):
            groups.append((f"l{l}.t2i.q_pe", self._x_w[l]["t2i.q"]))
            groups.append((f"l{l}.i2t.kv_pe", torch.cat((self._x_w[l]["i2t.k"], zeros(128)), 0)))
        groups.append(("final.q_pe", fa["q_proj"][0]))
        self.pe_off, off = {}, 0
        for name, w in groups:
            self.pe_off[name] = (off, w.shape[0])
            off += w.shape[0]
        self._pack("token_pe", _lin(torch.cat([w for _, w in groups], 0)), torch.zeros(off))
        self.const["out_tokens"] = tokens[:6].float().contiguous().to(dev)
        self._pack("final.q", _lin(fa["q_proj"][0]), fa["q_proj"][1])
        self.const["final.q_pe"] = cres(tokens @ fa["q_proj"][0].t())
        self._pack("final.kv", _lin(torch.cat((fa["k_proj"][0], fa["v_proj"][0]), 0)), torch.cat((fa["k_proj"][1], fa["v_proj"][1]), 0))
        self.const["final.kv_pe"] = cres(torch.cat((kpe @ fa["k_proj"][0].t(), torch.zeros(kpe.shape[0], 128)), 1))
        self._norm("norm_final", f"{D}.transformer.norm_final_attn", 256)
        # upscaling: ConvTranspose2d(k=2, s=2) weights [Cin, Cout, 2, 2] -> GEMM rows n = (dy*2+dx)*Cout + co
        for key, mod, cin, cout in (("up1", f"{D}.output_upscaling.0", 256, 64), ("up2", f"{D}.output_upscaling.3", 64, 32)):
            w = self.p.weight(mod, (cin, cout, 2, 2))
            b = self.p.bias(mod, cout)
            self._pack(key, _lin(w.permute(2, 3, 1, 0).reshape(4 * cout, cin)), b.repeat(4))
        self._norm("up_ln", f"{D}.output_upscaling.1", 64)
        for i in range(4):
            for j, (co, ci) in enumerate(((256, 256), (256, 256), (32, 256))):
                self._linear(f"hyper{i}.{j}", f"{D}.output_hypernetworks_mlps.{i}.layers.{j}", co, ci)
        for j, (co, ci) in enumerate(((256, 256), (256, 256), (4, 256))):
            self._linear(f"iou.{j}", f"{D}.iou_prediction_head.layers.{j}", co, ci)
        for j, (co, ci) in enumerate(((256, 256), (256, 256), (1, 256))):     # pred_obj_score_head: loaded (strictness), unused by the wrapper
            self.p.weight(f"{D}.pred_obj_score_head.layers.{j}", (co, ci)); self.p.bias(f"{D}.pred_obj_score_head.layers.{j}", co)
        # host-side intermediates of the constant folding above: no forward pass reads them, and a rank that receives its weights by
        # broadcast never has them (tests/test_distributed_cpu.py checks that everything left on this object is in the broadcast)
        del self.key_pe, self.sparse, self._sa_w, self._x_w, self.tokens

    def _refinement(self):
        if not self.use_refinement:
            self.refine_params = None
            return
        parts = []
        for j, k in enumerate(self.kernels):
            parts.append(self.p.tensor(f"refinement_layer.conv_branches.{j}.weight", (4, 1, k, k), "refine").reshape(-1))
            parts.append(self.p.tensor(f"refinement_layer.conv_branches.{j}.bias", (4,), "refine"))
        parts.append(self.p.tensor("refinement_layer.combiner_conv.weight", (1, 4 * len(self.kernels), 1, 1), "refine").reshape(-1))
        parts.append(self.p.tensor("refinement_layer.combiner_conv.bias", (1,), "refine"))
        self.refine_params = torch.cat([t.float() for t in parts]).to(self.device)


class Sam2Plan:
    """Launch plan for B images: x [B, R, R, 3] (NHWC, normalised) -> high_res, low_res, iou.

    prompts = 0: the reference's learned-prompt wrapper forward (one mask per image, sam2_infer.py:220-275).
    prompts = P > 0: upstream prompting, P prompts of `points` labelled points per image (a box = 2 corners + 1
    padding point); the caller fills `coords` [B*P, points, 2] / `labels` [B*P, points] before each run and the decoder
    runs on B*P (image, prompt) pairs, image-major (MaskDecoder repeat_image=True); outputs are [B*P, 1, ...]."""

    def __init__(self, wt, B, stream, dynamic_multimask_via_stability=True, prompts=0, points=3, high_res=True, attn="16"):
        """attn = "fp8" (16-bit plans only; BASELINE configs[4]): the AV contraction of Hiera's 256-key windows and global blocks runs on the
        block-scaled fp8 MFMA (cvmi_attn_desc.av_fp8); "16": operands in the plan's 16-bit type (default)."""
        self.wt, self.B, self.dt, self.dev = wt, B, wt.dtype, wt.device
        if attn not in ("16", "fp8") or (attn == "fp8" and not is16(wt.dtype)):
            raise ValueError("attn must be '16' or 'fp8' (fp8 needs an fp16 / bf16 plan)")
        self.av_fp8 = 1 if attn == "fp8" else 0
        self._ln1_stats, self._ln1_parts = None, 0                    # LayerNorm statistics handed from a block's MLP to the next block's norm1
        self.P, self.K, self.want_high_res = prompts, points, high_res
        if prompts and not wt.prompt_ok:
            raise _lib.CvmiError("this checkpoint carries no sam_prompt_encoder tensors: box / point prompts are unavailable")
        self.plan = Plan(stream)
        self.pool = {}
        self.dynamic = dynamic_multimask_via_stability
        self.act_bytes = 0
        self._build()
        torch.cuda.synchronize()

    def buf(self, H, W, C, dtype=None, tag=None, zero=False, batch=None):
        """Scratch buffers are shared between blocks of equal shape (launches are stream-ordered)."""
        dtype = self.dt if dtype is None else dtype
        batch = self.B if batch is None else batch
        key = (batch, H, W, C, dtype, tag)
        if tag is None or key not in self.pool:
            b = Buf(batch, H, W, C, dtype, self.dev, zero=zero)
            self.act_bytes += b.nbytes
            if tag is None:
                return b
            self.pool[key] = b
        return self.pool[key]

    def gemm(self, label, key, src, dst, act=ACT_NONE, res=None, kind="gemm", **kw):
        srcs = src if isinstance(src, list) else [(src, 0)]
        op_conv(self.plan, label, self.wt.pc[key], srcs, dst, act=act, res=res, kind=kind, **kw)

    def _build(self):
        wt, B = self.wt, self.B
        R = wt.image_size
        g = R // 4
        self.x_in = Buf(B, R, R, 3, self.dt, self.dev, zero=True)
        E = wt.hiera["embed_dim"]
        x = self.buf(g, g, E, F32)
        pos = wt.const["pos_embed"]
        self.plan.keep.append(pos)
        xs = Buf(B, g, g, 48, self.dt, self.dev)
        lib = _lib.load()
        op_call(self.plan, "s2d4", "stem", lib.cvmi_space_to_depth4, (self.x_in.t.data_ptr(), xs.t.data_ptr(), B, R, R, self.dt),
                keep=(self.x_in, xs), bytes_=2 * self.x_in.nbytes)
        op_conv(self.plan, "patch_embed", wt.pc["patch_embed"], [(xs.view(), 0)], x.view(), stride=1, pad=1, out_hw=(g, g),
                res=_ConstView(pos, E), res_mod=g * g, kind="stem")
        H = W = g
        stage_out = []
        for i, blk in enumerate(wt.blocks):
            x, H, W = self._block(i, blk, x, H, W)
            if i in wt.stage_ends:
                stage_out.append(x)
        self.stage_out = stage_out
        self._neck_decoder(stage_out)

    # ---- one Hiera block ---------------------------------------------------------------------------------
    def _block(self, i, blk, x, H, W):
        wt, B = self.wt, self.B
        dim, dout, heads, ws = blk["dim"], blk["dim_out"], blk["heads"], blk["window"]
        hd = dout // heads
        gam, bet = wt.ln[f"b{i}.norm1"]
        # Hiera pads the NORMALISED tokens up to a window multiple (zeros), attends inside the padded windows and crops
        # after the un-partition: the padded grid is a zero-initialised buffer whose valid region the LayerNorm writes
        Hp, Wp = (H, W) if ws == 0 else (-(-H // ws) * ws, -(-W // ws) * ws)
        padded = (Hp, Wp) != (H, W)
        if blk["q_pool"] and (ws == 0 or ws % 2 or Hp % 2 or Wp % 2):
            raise NotImplementedError("q-pool block needs an even window")
        tok_rows = not padded and (B * H * W) % 256 == 0          # token-stationary linears (tok_linear.hip) take 256-row workgroups
        tok_qkv = tok_rows and f"b{i}.qkv" in wt.tl                # ... and fuse norm1 into the qkv projection
        if dim != dout and os.environ.get("CVMI_SAM_QKVTOK_TRANS", "1") == "0":
            tok_qkv = False                                        # (A/B: the transition blocks' qkv through the tiled GEMM + a norm1 launch)
        tok_pool = tok_rows and dim != dout and f"b{i}.dimproj" in wt.tl and os.environ.get("CVMI_SAM_POOLFUSE", "1") != "0"     # norm1 + dimproj + 2 x 2 max-pool in one launch
        if padded:
            xn = self.buf(Hp, Wp, dim, tag="xn_padded", zero=True)
            op_layernorm(self.plan, f"b{i}.norm1", x.view(), gam, bet, xn.view(), 1e-6, pad=(H, W, Hp, Wp))
        elif not tok_qkv or (dim != dout and not tok_pool):
            xn = self.buf(H, W, dim, tag="xn")
            op_layernorm(self.plan, f"b{i}.norm1", x.view(), gam, bet, xn.view(), 1e-6)
        parts = self._ln1_parts
        ln1_stats = self._ln1_stats if (self._ln1_stats is not None and self._ln1_stats.numel() == 2 * B * H * W * max(parts, 1)) else None
        self._ln1_stats, self._ln1_parts = None, 0                 # (written by the previous block's MLP for exactly these rows)
        if dim != dout:
            short = self.buf(H // 2, W // 2, dout, F32)
            if tok_pool:
                op_tok_linear_pool(self.plan, f"b{i}.dimproj_pool", wt.tl[f"b{i}.dimproj"], x.view(), short.view(), (gam, bet, 1e-6),
                                   stats_in=ln1_stats if parts == 0 else None)
            else:
                pj = self.buf(H, W, dout, F32, tag="dimproj")
                self.gemm(f"b{i}.dimproj", f"b{i}.dimproj", xn.view(), pj.view(), out_hw=(H, W) if padded else None)
                op_maxpool2(self.plan, f"b{i}.pool", pj.view(), short.view())
        else:
            short = x
        qkv = self.buf(Hp, Wp, 3 * dout, tag="qkv")
        if tok_qkv:
            op_tok_linear(self.plan, f"b{i}.qkv", wt.tl[f"b{i}.qkv"], x.view(), qkv.view(), ln=(gam, bet, 1e-6), stats_in=ln1_stats, stats_parts=parts)
        else:
            self.gemm(f"b{i}.qkv", f"b{i}.qkv", xn.view(), qkv.view())
        OH, OW = (H // 2, W // 2) if blk["q_pool"] else (H, W)
        OHp, OWp = (Hp // 2, Wp // 2) if blk["q_pool"] else (Hp, Wp)
        ao = self.buf(OHp, OWp, dout, tag="ao")
        es = ESIZE[self.dt]
        base = qkv.t.data_ptr()
        C3 = 3 * dout
        if ws > 0:
            nwin = B * (Hp // ws) * (Wp // ws)
            nq = (ws // 2) ** 2 if blk["q_pool"] else ws * ws
            desc = make_attn_desc(q=base, k=base + dout * es, v=base + 2 * dout * es, o=ao.t.data_ptr(),
                                  q_sb=0, q_sh=hd, q_st=C3, k_sb=0, k_sh=hd, k_st=C3, v_sb=0, v_sh=hd, v_st=C3,
                                  o_sb=0, o_sh=hd, o_st=dout, B=nwin, heads=heads, Nq=nq, Nk=ws * ws, dqk=hd, dv=hd,
                                  scale=hd ** -0.5, dtype=self.dt, win=ws, grid_h=Hp, grid_w=Wp, q_pool=1 if blk["q_pool"] else 0, av_fp8=self.av_fp8,
                                  q_log2=1 if wt.q_log2 else 0)
            fl = 4 * nwin * heads * nq * ws * ws * hd
        else:
            N = H * W
            desc = make_attn_desc(q=base, k=base + dout * es, v=base + 2 * dout * es, o=ao.t.data_ptr(),
                                  q_sb=N * C3, q_sh=hd, q_st=C3, k_sb=N * C3, k_sh=hd, k_st=C3, v_sb=N * C3, v_sh=hd, v_st=C3,
                                  o_sb=N * dout, o_sh=hd, o_st=dout, B=B, heads=heads, Nq=N, Nk=N, dqk=hd, dv=hd,
                                  scale=hd ** -0.5, dtype=self.dt, win=0, grid_h=0, grid_w=0, q_pool=0, av_fp8=self.av_fp8, q_log2=1 if wt.q_log2 else 0)
            fl = 4 * B * heads * N * N * hd
        op_attention(self.plan, f"b{i}.attn", desc, (qkv, ao), bytes_=qkv.nbytes + ao.nbytes, flops=fl)
        self.plan.ops[-1] = (self.plan.ops[-1][0], "attn_global" if ws == 0 else "attn_window") + self.plan.ops[-1][2:]
        # x = shortcut + proj(attn)   (in place on the f32 residual stream)
        tok_out = not padded and (B * OH * OW) % 256 == 0
        # norm2's statistics travel from the launch that writes x to the launch that normalises it (fc1 then reads x once, not twice)
        fwd = (tok_out and f"b{i}.proj" in wt.tl and f"b{i}" not in wt.mlp and f"b{i}.fc1" in wt.tl
               and os.environ.get("CVMI_SAM_LNSTATS", "1") != "0")
        sparts = tok_linear_stats_parts(B * OH * OW, dout, dout) if fwd else 0     # (a launch with fewer row blocks than CUs writes its statistics in slices)
        stats = torch.empty(B * OH * OW, max(sparts, 1), 2, dtype=torch.float32, device=self.dev) if fwd else None
        if tok_out and f"b{i}.proj" in wt.tl:
            op_tok_linear(self.plan, f"b{i}.proj", wt.tl[f"b{i}.proj"], ao.view(), short.view(), residual=True, stats_out=stats, stats_eps=1e-6)
        else:
            self.gemm(f"b{i}.proj", f"b{i}.proj", ao.view(), short.view(), res=short.view(), out_hw=(OH, OW) if padded else None)
        x = short
        gam, bet = wt.ln[f"b{i}.norm2"]
        if f"b{i}" in wt.mlp:
            nxt = wt.blocks[i + 1] if i + 1 < len(wt.blocks) else None
            if nxt is not None and os.environ.get("CVMI_SAM_LNSTATS", "1") != "0" and f"b{i + 1}.qkv" in wt.tl and (B * OH * OW) % 256 == 0:
                self._ln1_stats = torch.empty(B * OH * OW, 2, dtype=torch.float32, device=self.dev)      # the next block's norm1 reads x once
            op_hiera_mlp(self.plan, f"b{i}.mlp", wt.mlp[f"b{i}"], x.view(), gam, bet, 1e-6, stats_out=self._ln1_stats, stats_eps=1e-6)
        else:
            hid = self.buf(OH, OW, 4 * dout, tag="hid")
            if tok_out and f"b{i}.fc1" in wt.tl:                   # norm2 fused into fc1 (+ GELU)
                op_tok_linear(self.plan, f"b{i}.fc1", wt.tl[f"b{i}.fc1"], x.view(), hid.view(), ln=(gam, bet, 1e-6), act=ACT_GELU, stats_in=stats, stats_parts=sparts)
            else:
                xn2 = self.buf(OH, OW, dout, tag="xn")
                op_layernorm(self.plan, f"b{i}.norm2", x.view(), gam, bet, xn2.view(), 1e-6)
                self.gemm(f"b{i}.fc1", f"b{i}.fc1", xn2.view(), hid.view(), act=ACT_GELU)
            # a tiled GEMM hands the next block's norm1 raw per-slice sums (the 256 x 192 kernel: stage-3 fc2 shape); tok_linear adds them up
            rs = None
            nxt_qkv = f"b{i + 1}.qkv" in wt.tl and i + 1 < len(wt.blocks) and wt.blocks[i + 1]["dim"] == dout
            if (nxt_qkv and tok_out and self.dt != F32 and row_stats_supported(B * OH * OW, dout, 4 * dout)
                    and os.environ.get("CVMI_SAM_LNSTATS", "1") != "0" and os.environ.get("CVMI_SAM_LNSTATS_FC2", "1") != "0"):
                rs = torch.empty(B * OH * OW, dout // 96, 2, dtype=torch.float32, device=self.dev)
                self._ln1_stats, self._ln1_parts = rs, dout // 96
            self.gemm(f"b{i}.fc2", f"b{i}.fc2", hid.view(), x.view(), res=x.view(), row_stats=rs)
        if i in self.wt.stage_ends and i != len(self.wt.blocks) - 1:
            # the next block writes its own shortcut buffer (dim change) -> x stays intact as the stage output
            pass
        return x, OH, OW

    # ---- neck + mask decoder + tail -------------------------------------------------------------------------
    def _neck_decoder(self, st):
        wt, B, dt = self.wt, self.B, self.dt
        R = wt.image_size
        f0, f1, fs = R // 4, R // 8, R // 16
        # GEMM operands in compute dtype (stage outputs are f32 residual streams)
        def cast(buf, tag):
            if dt == F32:
                return buf
            o = self.buf(buf.H, buf.W, buf.C, dt)
            op_cast(self.plan, f"cast.{tag}", buf.view(), o.view())
            return o
        feat_s0 = self.buf(f0, f0, 32)
        feat_s1 = self.buf(f1, f1, 64)
        fused = os.environ.get("CVMI_SAM_NECKCAST", "1") != "0"
        for key, src, dstb, tag in (("feat_s0", st[0], feat_s0, "s0"), ("feat_s1", st[1], feat_s1, "s1")):
            if fused and key in wt.tl and src.dtype == F32 and (B * src.H * src.W) % 256 == 0:
                op_tok_linear(self.plan, key, wt.tl[key], src.view(), dstb.view(), ln="cast", kind="neck")     # reads the f32 stage output itself
            else:
                self.gemm(key, key, cast(src, tag).view(), dstb.view(), kind="neck")
        s2, s3 = cast(st[2], "s2"), cast(st[3], "s3")
        lib = _lib.load()
        NP = self.P                                   # prompts per image (0: learned prompts)
        NB = B * NP if NP else B                      # decoder batch: (image, prompt) pairs, image-major
        P = fs * fs
        bufd = lambda *a, **k: self.buf(*a, batch=NB, **k)
        if not NP:
            # src = FPN level 2 + learned dense prompt   (f32: it is the decoder's residual stream "keys")
            keys = self.buf(fs, fs, 256, F32)
            dense = wt.const["dense"]
            self.gemm("embed", "embed", [(s2.view(), 0), (s3.view(), 1)], keys.view(), res=_ConstView(dense, 256), res_mod=fs * fs, kind="neck")
            T = wt.n_tok
            # token stream (f32).  Layer 0 reads the constant tokens (qn0) and REPLACES the stream, so it needs no init.
            qn0 = Buf(B, 1, T, 256, dt, self.dev)
            qn0.t.copy_(wt.const["tokens"].to(TORCH_DTYPE[dt]).view(1, 1, T, 256).expand(B, 1, T, 256))
            tpe = lambda name, C_: dict(res=_ConstView(wt.const[name], C_), res_mod=T)
        else:
            # src = FPN level 2 + no_mem_embed + no_mask_embed (both in the GEMM bias), repeated for the prompts of each image
            emb = self.buf(fs, fs, 256, F32)
            self.gemm("embed", "embed_box", [(s2.view(), 0), (s3.view(), 1)], emb.view(), kind="neck")
            keys = bufd(fs, fs, 256, F32)
            # Until layer 0's image -> token attention writes into it, the image stream of every prompt of an image IS that image's
            # embedding: in fp16 mode layer 0 reads the B shared copies (k / v and q projections on B images instead of B * P,
            # attention kernels index the shared batch entry, the first residual add broadcasts) and the repeat pass disappears.
            # f32 parity mode keeps the literal repeat_image formulation (upstream MaskDecoder.predict_masks) as the cross-check.
            self.share_l0 = is16(dt) and os.environ.get("CVMI_SAM_SHARE_L0", "1") != "0"
            if not self.share_l0:
                op_call(self.plan, "repeat_embed", "decoder", lib.cvmi_repeat_images, (emb.t.data_ptr(), keys.t.data_ptr(), P * 256 * 4, B, NP),
                        keep=(emb, keys), bytes_=(B + NB) * P * 256 * 4)
            T = 6 + self.K
            self.coords = torch.zeros(NB, self.K, 2, dtype=torch.float32, device=self.dev)
            self.labels = torch.full((NB, self.K), -1, dtype=torch.int32, device=self.dev)
            tok0 = Buf(NB, 1, T, 256, F32, self.dev)
            qn0 = Buf(NB, 1, T, 256, dt, self.dev)
            op_call(self.plan, "prompt_tokens", "decoder", lib.cvmi_prompt_tokens,
                    (self.coords.data_ptr(), self.labels.data_ptr(), wt.const["gauss"].data_ptr(), wt.const["out_tokens"].data_ptr(),
                     wt.const["prompt_table"].data_ptr(), float(R), tok0.t.data_ptr(), qn0.t.data_ptr(), dt, NB, self.K, 6),
                    keep=(tok0, qn0), bytes_=NB * T * 256 * 6)
            npe = wt.pc["token_pe"].N
            pe_all = Buf(NB, 1, T, npe, dt, self.dev)
            self.gemm("token_pe", "token_pe", qn0.view(), pe_all.view(), kind="decoder")
            tpe = lambda name, C_: dict(res=pe_all.view(wt.pe_off[name][0], C_))
            self.tokens0 = tok0
        self.feat_s0, self.feat_s1, self.keys0 = feat_s0, feat_s1, keys
        q = Buf(NB, 1, T, 256, F32, self.dev, zero=True)

        def ln(label, key, src, dst, dst2=None):
            gam, bet = wt.ln[key]
            op_layernorm(self.plan, label, src.view(), gam, bet, dst.view(), 1e-5, dst2=dst2.view() if dst2 is not None else None)

        def attention(label, qb, q_off, kb, k_off, vb, v_off, ob, Nq, Nk, hd, q_bdiv=0, kv_bdiv=0):
            es = ESIZE[dt]
            desc = make_attn_desc(q=qb.t.data_ptr() + q_off * es, k=kb.t.data_ptr() + k_off * es, v=vb.t.data_ptr() + v_off * es, o=ob.t.data_ptr(),
                                  q_sb=Nq * qb.C, q_sh=hd, q_st=qb.C, k_sb=Nk * kb.C, k_sh=hd, k_st=kb.C, v_sb=Nk * vb.C, v_sh=hd, v_st=vb.C,
                                  o_sb=Nq * ob.C, o_sh=hd, o_st=ob.C, B=NB, heads=8, Nq=Nq, Nk=Nk, dqk=hd, dv=hd, scale=hd ** -0.5, dtype=dt,
                                  win=0, grid_h=0, grid_w=0, q_pool=0, q_bdiv=q_bdiv, kv_bdiv=kv_bdiv)
            op_attention(self.plan, label, desc, (qb, kb, vb, ob), flops=4 * NB * 8 * Nq * Nk * hd)
            self.plan.ops[-1] = (self.plan.ops[-1][0], "decoder") + self.plan.ops[-1][2:]

        dual = is16(dt)                          # norm4 writes the 16-bit operand copy of the image stream itself (no cast pass)
        qn = bufd(1, T, 256, tag="qn")           # compute-dtype copies of the f32 streams
        kn = bufd(fs, fs, 256, tag="kn")
        G = lambda *a, **k: self.gemm(*a, kind="decoder", **k)
        for l in (0, 1):
            p = f"l{l}"
            # --- self attention on the tokens
            qkv = bufd(1, T, 768, tag="sa_qkv")
            if l == 0:
                G(f"{p}.sa.qkv", f"{p}.sa.qkv", qn0.view(), qkv.view())
            else:
                op_cast(self.plan, f"{p}.sa.cast", q.view(), qn.view())
                G(f"{p}.sa.qkv", f"{p}.sa.qkv", qn.view(), qkv.view(), **tpe(f"{p}.sa.qkv_pe", 768))
            ao = bufd(1, T, 256, tag="sa_ao")
            attention(f"{p}.sa.attn", qkv, 0, qkv, 256, qkv, 512, ao, T, T, 32)
            if l == 0:
                G(f"{p}.sa.out", f"{p}.sa.out", ao.view(), q.view())                         # layer 0: queries REPLACED (skip_first_layer_pe)
            else:
                G(f"{p}.sa.out", f"{p}.sa.out", ao.view(), q.view(), res=q.view())
            ln(f"{p}.norm1", f"{p}.norm1", q, q)
            # --- tokens attend to the image
            op_cast(self.plan, f"{p}.t2i.castq", q.view(), qn.view())
            share = l == 0 and NP and getattr(self, "share_l0", False)      # layer 0 of the box path: image-side tensors per IMAGE
            if share:
                kn_l = self.buf(fs, fs, 256, tag="kn_shared")
                op_cast(self.plan, f"{p}.t2i.castk", emb.view(), kn_l.view())
            else:
                kn_l = kn
                if l == 0 or not dual:
                    op_cast(self.plan, f"{p}.t2i.castk", keys.view(), kn.view())
            tq = bufd(1, T, 128, tag="t2i_q")
            G(f"{p}.t2i.q", f"{p}.t2i.q", qn.view(), tq.view(), **tpe(f"{p}.t2i.q_pe", 128))
            kv = self.buf(fs, fs, 256, tag="t2i_kv_shared") if share else bufd(fs, fs, 256, tag="t2i_kv")
            G(f"{p}.t2i.kv", f"{p}.t2i.kv", kn_l.view(), kv.view(), res=_ConstView(wt.const[f"{p}.t2i.kv_pe"], 256), res_mod=P)
            ao2 = bufd(1, T, 128, tag="t2i_ao")
            attention(f"{p}.t2i.attn", tq, 0, kv, 0, kv, 128, ao2, T, P, 16, kv_bdiv=NP if share else 0)
            G(f"{p}.t2i.out", f"{p}.t2i.out", ao2.view(), q.view(), res=q.view())
            ln(f"{p}.norm2", f"{p}.norm2", q, q)
            # --- MLP on the tokens
            op_cast(self.plan, f"{p}.mlp.cast", q.view(), qn.view())
            hid = bufd(1, T, 2048, tag="mlp_hid")
            G(f"{p}.mlp1", f"{p}.mlp1", qn.view(), hid.view(), act=ACT_RELU)
            G(f"{p}.mlp2", f"{p}.mlp2", hid.view(), q.view(), res=q.view())
            ln(f"{p}.norm3", f"{p}.norm3", q, q)
            # --- image attends to the tokens
            op_cast(self.plan, f"{p}.i2t.castq", q.view(), qn.view())
            iq = self.buf(fs, fs, 128, tag="i2t_q_shared") if share else bufd(fs, fs, 128, tag="i2t_q")
            G(f"{p}.i2t.q", f"{p}.i2t.q", kn_l.view(), iq.view(), res=_ConstView(wt.const[f"{p}.i2t.q_pe"], 128), res_mod=P)
            ikv = bufd(1, T, 256, tag="i2t_kv")
            G(f"{p}.i2t.kv", f"{p}.i2t.kv", qn.view(), ikv.view(), **tpe(f"{p}.i2t.kv_pe", 256))
            ao3 = bufd(fs, fs, 128, tag="i2t_ao")
            attention(f"{p}.i2t.attn", iq, 0, ikv, 0, ikv, 128, ao3, P, T, 16, q_bdiv=NP if share else 0)
            if share:      # first write of the per-prompt image stream: keys[b] = emb[b / P] + out-projection
                G(f"{p}.i2t.out", f"{p}.i2t.out", ao3.view(), keys.view(), res=emb.view(), res_mod=P, res_rep=NP)
            else:
                G(f"{p}.i2t.out", f"{p}.i2t.out", ao3.view(), keys.view(), res=keys.view())
            ln(f"{p}.norm4", f"{p}.norm4", keys, keys, dst2=kn if dual else None)      # + the fp16 copy the next k / v projection reads
        # --- final token -> image attention
        op_cast(self.plan, "final.castq", q.view(), qn.view())
        if not dual:
            op_cast(self.plan, "final.castk", keys.view(), kn.view())
        tq = bufd(1, T, 128, tag="t2i_q")
        G("final.q", "final.q", qn.view(), tq.view(), **tpe("final.q_pe", 128))
        kv = bufd(fs, fs, 256, tag="t2i_kv")
        G("final.kv", "final.kv", kn.view(), kv.view(), res=_ConstView(wt.const["final.kv_pe"], 256), res_mod=P)
        ao2 = bufd(1, T, 128, tag="t2i_ao")
        attention("final.attn", tq, 0, kv, 0, kv, 128, ao2, T, P, 16)
        G("final.out", "final.out", ao2.view(), q.view(), res=q.view())
        ln("norm_final", "norm_final", q, q)
        self.tokens_out, self.keys_out = q, keys
        # --- upscaling: act1(ln1(dc1(src) + s1)) ; act2(dc2(.) + s0)
        u1 = bufd(f1, f1, 64)
        G("up1", "up1", kn.view(), u1.view(), res=feat_s1.view(), shuffle_cout=64, res_rep=NP)
        gam, bet = wt.ln["up_ln"]
        op_layernorm(self.plan, "up_ln", u1.view(), gam, bet, u1.view(), 1e-6, act=ACT_GELU)
        u2 = bufd(f0, f0, 32)
        G("up2", "up2", u1.view(), u2.view(), res=feat_s0.view(), shuffle_cout=32, act=ACT_GELU, act_after_res=True, res_rep=NP)
        self.up = u2
        # --- hypernetwork MLPs on the 4 mask tokens, IoU head on the iou token (rows strided by T*256)
        op_cast(self.plan, "heads.cast", q.view(), qn.view())
        hyper = Buf(NB, 1, 4, 32, F32, self.dev)
        for i in range(4):
            src = _RowsView(qn.t, NB, 256, ld=T * 256, offset=(2 + i) * 256, dtype=dt)
            h1 = bufd(1, 1, 256, tag="h1"); h2 = bufd(1, 1, 256, tag="h2")
            G(f"hyper{i}.0", f"hyper{i}.0", src, h1.view(), act=ACT_RELU)
            G(f"hyper{i}.1", f"hyper{i}.1", h1.view(), h2.view(), act=ACT_RELU)
            G(f"hyper{i}.2", f"hyper{i}.2", h2.view(), _RowsView(hyper.t, NB, 32, ld=128, offset=i * 32, dtype=F32))
        iou4 = Buf(NB, 1, 1, 4, F32, self.dev)
        src = _RowsView(qn.t, NB, 256, ld=T * 256, offset=256, dtype=dt)
        h1 = bufd(1, 1, 256, tag="h1"); h2 = bufd(1, 1, 256, tag="h2")
        G("iou.0", "iou.0", src, h1.view(), act=ACT_RELU)
        G("iou.1", "iou.1", h1.view(), h2.view(), act=ACT_RELU)
        G("iou.2", "iou.2", h2.view(), iou4.view(), act=ACT_SIGMOID)
        self.hyper, self.iou4 = hyper, iou4
        # --- masks, dynamic multimask selection, upsample + refinement
        P0 = f0 * f0
        self.masks4 = torch.empty(NB, 4, f0, f0, dtype=torch.float32, device=self.dev)
        self.areas = torch.zeros(NB, 2, dtype=torch.int32, device=self.dev)
        self.low_res = torch.empty(NB, 1, f0, f0, dtype=torch.float32, device=self.dev)
        self.iou = torch.empty(NB, 1, dtype=torch.float32, device=self.dev)
        self.sel = torch.zeros(NB, dtype=torch.int32, device=self.dev)
        op_call(self.plan, "hyper_masks", "tail", lib.cvmi_hyper_masks,
                (hyper.t.data_ptr(), 32, u2.t.data_ptr(), 32, dt, 32, self.masks4.data_ptr(), self.areas.data_ptr(), NB, P0, 0.05),
                keep=(hyper, u2), bytes_=NB * P0 * (32 * ESIZE[dt] + 16), flops=2 * NB * P0 * 128)
        op_call(self.plan, "select_mask", "tail", lib.cvmi_select_mask,
                (self.masks4.data_ptr(), self.areas.data_ptr(), iou4.t.data_ptr(), 4, 1 if self.dynamic else 0, 0.98, self.low_res.data_ptr(),
                 self.iou.data_ptr(), self.sel.data_ptr(), NB, P0), bytes_=NB * P0 * 8)
        if not self.want_high_res:
            self.high_res = None
            return
        self.high_res = torch.empty(NB, 1, R, R, dtype=torch.float32, device=self.dev)
        if wt.refine_params is not None and not NP:            # the refinement head belongs to the learned-prompt wrapper (sam2_infer.py:269-270)
            import ctypes as C
            ks = (C.c_int * len(wt.kernels))(*wt.kernels)
            taps = sum(k * k for k in wt.kernels)
            op_call(self.plan, "upsample_refine", "tail", lib.cvmi_upsample_refine,
                    (self.low_res.data_ptr(), NB, f0, f0, self.high_res.data_ptr(), R, R, wt.refine_params.data_ptr(), ks, len(wt.kernels), 4),
                    keep=(ks,), bytes_=NB * (P0 + R * R) * 4, flops=2 * NB * R * R * 4 * taps)
        else:
            op_call(self.plan, "upsample", "tail", lib.cvmi_bilinear_f32,
                    (self.low_res.data_ptr(), NB, f0, f0, self.high_res.data_ptr(), R, R, None, 0.0), bytes_=NB * (P0 + R * R) * 4)

class _ConstView:
    """Constant [rows, C] device tensor posing as a residual View (ptr, ld)."""

    def __init__(self, t, C):
        self.t, self.c = t, C

    ptr = property(lambda s: s.t.data_ptr())
    ld = property(lambda s: s.c)


class _RowsView:
    """Strided row matrix posing as a [B,1,1,C] conv source / destination."""

    def __init__(self, t, rows, C, ld, offset, dtype):
        self.t, self.B, self.H, self.W, self.c, self._ld, self.off, self.dtype = t, rows, 1, 1, C, ld, offset, dtype

    ptr = property(lambda s: s.t.data_ptr() + s.off * s.t.element_size())
    ld = property(lambda s: s._ld)
