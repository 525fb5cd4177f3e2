"""Oracle: SAM 2.1 image path (Hiera trunk, FPN neck, prompt-encoder dense PE, mask decoder) and the
reference's learned-prompt wrapper, CPU fp32, plain PyTorch.  TEST INFRASTRUCTURE ONLY.

Follows, in the reference:
  * src/sam2_infer.py:191-275   SAM2ImageWrapper (dataflow, learned prompts, refinement call)
  * src/sam2_infer.py:130-189   MultiKernelRefinement         (PINNED: tests/golden/refinement.npz)
  * src/sam2_infer.py:88-128    SAM2Transforms.postprocess_masks (PINNED: tests/golden/postprocess.npz)
  * src/sam2_infer.py:29-56     SAM2Transforms.__call__ (ToTensor / Resize / Normalize)
  * src/circuit_analyzer.py:156-223  LoRA target list and factory kwargs (r=4, alpha=16)
  * models/configs/sam2.1_hiera_l.yaml   hyper-parameters
The network itself (Hiera, FpnNeck, PromptEncoder, MaskDecoder, TwoWayTransformer) lives in the
un-vendored facebookresearch/sam2 package (requirements.txt:12, unpinned HEAD) and LoRA in `peft`
(requirements.txt:13): restated here from the published algorithm (SURVEY.md 8(a) rows B3-B13,
Table H) with upstream parameter names, and cross-checked on CPU against the independent
`transformers` SAM2 implementation (tests/test_oracle_sam2_cpu.py).  parity unpinned by the
reference itself: it holds no tests or fixtures for the network.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

LORA_R, LORA_ALPHA = 4, 16


# ---- LoRA-wrapped layers with peft's parameter names -------------------------------------------------
class LoRALinear(nn.Module):
    """y = W x + b + (alpha/r) * B(A(x)); keys: base_layer.{weight,bias}, lora_A.default.weight, lora_B.default.weight"""

    def __init__(self, cin, cout, r=LORA_R, alpha=LORA_ALPHA):
        super().__init__()
        self.base_layer = nn.Linear(cin, cout)
        self.lora_A = nn.ModuleDict({"default": nn.Linear(cin, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, cout, bias=False)})
        self.scaling = alpha / r

    def forward(self, x):
        return self.base_layer(x) + self.scaling * self.lora_B["default"](self.lora_A["default"](x))


class LoRAConv1x1(nn.Module):
    def __init__(self, cin, cout, r=LORA_R, alpha=LORA_ALPHA):
        super().__init__()
        self.base_layer = nn.Conv2d(cin, cout, 1)
        self.lora_A = nn.ModuleDict({"default": nn.Conv2d(cin, r, 1, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Conv2d(r, cout, 1, bias=False)})
        self.scaling = alpha / r

    def forward(self, x):
        return self.base_layer(x) + self.scaling * self.lora_B["default"](self.lora_A["default"](x))


def _linear(cin, cout, lora):
    return LoRALinear(cin, cout) if lora else nn.Linear(cin, cout)


class MLP(nn.Module):
    def __init__(self, cin, hidden, cout, num_layers, act=nn.ReLU, sigmoid_output=False, lora=()):
        super().__init__()
        dims = [cin] + [hidden] * (num_layers - 1) + [cout]
        self.layers = nn.ModuleList(_linear(dims[i], dims[i + 1], i in lora) for i in range(num_layers))
        self.act = act()
        self.sigmoid_output = sigmoid_output

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = self.act(layer(x)) if i < len(self.layers) - 1 else layer(x)
        return torch.sigmoid(x) if self.sigmoid_output else x


# ---- Hiera trunk ----------------------------------------------------------------------------------------
def do_pool(x, stride):
    return F.max_pool2d(x.permute(0, 3, 1, 2), stride, stride).permute(0, 2, 3, 1)


def window_partition(x, ws):
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


class MultiScaleAttention(nn.Module):
    def __init__(self, dim, dim_out, heads, q_pool, lora_qkv=False):
        super().__init__()
        self.heads, self.q_pool = heads, q_pool
        self.qkv = _linear(dim, dim_out * 3, lora_qkv)
        self.proj = nn.Linear(dim_out, dim_out)

    def forward(self, x):
        B, H, W, _ = x.shape
        qkv = self.qkv(x).reshape(B, H * W, 3, self.heads, -1)
        q, k, v = torch.unbind(qkv, 2)
        if self.q_pool:
            q = do_pool(q.reshape(B, H, W, -1), self.q_pool)
            H, W = q.shape[1:3]
            q = q.reshape(B, H * W, self.heads, -1)
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), -1) @ v
        return self.proj(a.transpose(1, 2).reshape(B, H, W, -1))


class MultiScaleBlock(nn.Module):
    def __init__(self, dim, dim_out, heads, q_stride, window, lora=()):
        super().__init__()
        self.dim, self.dim_out, self.window, self.q_stride = dim, dim_out, window, q_stride
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = MultiScaleAttention(dim, dim_out, heads, q_stride, "attn.qkv" in lora)
        self.norm2 = nn.LayerNorm(dim_out, eps=1e-6)
        self.mlp = MLP(dim_out, dim_out * 4, dim_out, 2, act=nn.GELU, lora=(0,) if "mlp.layers.0" in lora else ())
        if dim != dim_out:
            self.proj = _linear(dim, dim_out, "proj" in lora)

    def forward(self, x):
        shortcut = x
        x = self.norm1(x)
        if self.dim != self.dim_out:
            shortcut = do_pool(self.proj(x), self.q_stride)
        ws = self.window
        if ws > 0:
            H, W = x.shape[1:3]
            x, pad_hw = window_partition(x, ws)
        x = self.attn(x)
        if self.q_stride:
            ws = self.window // self.q_stride
            H, W = shortcut.shape[1:3]
            pad_hw = (H + (ws - H % ws) % ws, W + (ws - W % ws) % ws)
        if self.window > 0:
            x = window_unpartition(x, ws, pad_hw, (H, W))
        x = shortcut + x
        return x + self.mlp(self.norm2(x))


class Hiera(nn.Module):
    def __init__(self, embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43),
                 window_spec=(8, 4, 16, 8), bkg_size=(7, 7), q_pool=3, q_stride=2, lora_blocks=None):
        super().__init__()
        lora_blocks = lora_blocks or {}
        depth = sum(stages)
        self.stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
        q_pool_blocks = [x + 1 for x in self.stage_ends[:-1]][:q_pool]
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, embed_dim, 7, 4, 3)
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, *bkg_size))
        self.pos_embed_window = nn.Parameter(torch.zeros(1, embed_dim, window_spec[0], window_spec[0]))
        self.blocks = nn.ModuleList()
        cur, dim, heads = 1, embed_dim, num_heads
        self.channel_list = []
        for i in range(depth):
            dim_out = dim
            window = window_spec[cur - 1]
            if i in global_att_blocks:
                window = 0
            if i - 1 in self.stage_ends:
                dim_out, heads, cur = dim * 2, heads * 2, cur + 1
            self.blocks.append(MultiScaleBlock(dim, dim_out, heads, q_stride if i in q_pool_blocks else None, window,
                                               lora_blocks.get(i, ())))
            dim = dim_out
            if i in self.stage_ends:
                self.channel_list.append(dim)

    def pos(self, hw):
        pe = F.interpolate(self.pos_embed, size=hw, mode="bicubic")
        win = self.pos_embed_window
        pe = pe + win.tile([x // y for x, y in zip(pe.shape, win.shape)])
        return pe.permute(0, 2, 3, 1)

    def forward(self, x):
        x = self.patch_embed.proj(x).permute(0, 2, 3, 1)
        x = x + self.pos(x.shape[1:3])
        outs = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in self.stage_ends:
                outs.append(x.permute(0, 3, 1, 2))
        return outs


class FpnNeck(nn.Module):
    def __init__(self, channel_list, d_model=256, top_down=(2, 3), lora_convs=()):
        super().__init__()
        self.convs = nn.ModuleList()
        for j, c in enumerate(channel_list):          # channel_list is deepest-first: [1152, 576, 288, 144]
            m = nn.Module()
            m.conv = LoRAConv1x1(c, d_model) if j in lora_convs else nn.Conv2d(c, d_model, 1)
            self.convs.append(m)
        self.top_down = top_down

    def forward(self, xs):
        n = len(self.convs) - 1
        out, prev = [None] * len(xs), None
        for i in range(n, -1, -1):
            lat = self.convs[n - i].conv(xs[i])
            if i in self.top_down and prev is not None:
                prev = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
            else:
                prev = lat
            out[i] = prev
        return out


# ---- prompt encoder's dense positional encoding --------------------------------------------------------------
def dense_pe(gaussian, h, w):
    """PositionEmbeddingRandom(size): grid centres -> [-1,1] -> @G -> 2pi -> [sin,cos]; [1, 2*G.shape[1], h, w]."""
    y = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    x = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    c = 2 * torch.stack((xx, yy), -1) - 1
    c = 2 * math.pi * (c @ gaussian)
    return torch.cat((torch.sin(c), torch.cos(c)), -1).permute(2, 0, 1).unsqueeze(0)


class PromptEncoder(nn.Module):
    """Upstream sam2 PromptEncoder, the part box / point prompting uses (no mask input): random-Fourier
    encoding of pixel-centre coordinates plus one learned vector per label.  Labels: -1 padding ("not a point"),
    0 / 1 negative / positive click, 2 / 3 box top-left / bottom-right corner.  The reference never calls this
    (its prompts are learned constants, src/sam2_infer.py:206-209); it serves `infer_masks(images, boxes)`."""

    def __init__(self, dim=256, image_size=1024):
        super().__init__()
        self.image_size = image_size
        self.pe_layer = nn.Module()
        self.pe_layer.register_buffer("positional_encoding_gaussian_matrix", torch.randn(2, dim // 2))
        self.point_embeddings = nn.ModuleList(nn.Embedding(1, dim) for _ in range(4))
        self.not_a_point_embed = nn.Embedding(1, dim)
        self.no_mask_embed = nn.Embedding(1, dim)

    def encode_coords(self, coords):
        """coords [..., 2] (x, y) in input pixels -> [..., dim]."""
        c = 2 * (coords / self.image_size) - 1
        c = 2 * math.pi * (c @ self.pe_layer.positional_encoding_gaussian_matrix)
        return torch.cat((torch.sin(c), torch.cos(c)), -1)

    def embed_points(self, coords, labels, pad):
        """coords [P, K, 2], labels [P, K] -> sparse [P, K (+1), dim]."""
        coords = coords + 0.5                                            # pixel centre
        if pad:
            coords = torch.cat((coords, torch.zeros(coords.shape[0], 1, 2)), 1)
            labels = torch.cat((labels, -torch.ones(labels.shape[0], 1, dtype=labels.dtype)), 1)
        e = self.encode_coords(coords)
        e = torch.where((labels == -1)[..., None], torch.zeros_like(e), e)
        e = e + torch.where((labels == -1)[..., None], self.not_a_point_embed.weight, torch.zeros_like(e))
        for i in range(4):
            e = e + torch.where((labels == i)[..., None], self.point_embeddings[i].weight, torch.zeros_like(e))
        return e

    def embed_boxes(self, boxes):
        """boxes [P, 4] xyxy -> [P, 3, dim]: the predictor hands boxes over as two labelled points (2, 3) + padding."""
        labels = torch.tensor([[2, 3]], dtype=torch.long).expand(boxes.shape[0], 2)
        return self.embed_points(boxes.reshape(-1, 2, 2), labels, pad=True)

    def dense_no_mask(self, fs):
        return self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(1, -1, fs, fs)


# ---- mask decoder ------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, dim, heads, downsample=1, lora=()):
        super().__init__()
        inner = dim // downsample
        self.heads = heads
        self.q_proj = _linear(dim, inner, "q_proj" in lora)
        self.k_proj = _linear(dim, inner, "k_proj" in lora)
        self.v_proj = _linear(dim, inner, "v_proj" in lora)
        self.out_proj = _linear(inner, dim, "out_proj" in lora)

    def forward(self, q, k, v):
        q, k, v = self.q_proj(q), self.k_proj(k), self.v_proj(v)

        def split(t):
            b, n, c = t.shape
            return t.reshape(b, n, self.heads, c // self.heads).transpose(1, 2)
        q, k, v = split(q), split(k), split(v)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), -1) @ v
        b, h, n, d = a.shape
        return self.out_proj(a.transpose(1, 2).reshape(b, n, h * d))


class TwoWayBlock(nn.Module):
    def __init__(self, dim, heads, mlp_dim, skip_first_layer_pe, lora=True):
        super().__init__()
        full = ("q_proj", "k_proj", "v_proj", "out_proj") if lora else ()
        qkv = ("q_proj", "k_proj", "v_proj") if lora else ()
        self.self_attn = Attention(dim, heads, 1, full)
        self.norm1 = nn.LayerNorm(dim)
        self.cross_attn_token_to_image = Attention(dim, heads, 2, full)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, mlp_dim, dim, 2, act=nn.ReLU, lora=(0, 1) if lora else ())
        self.norm3 = nn.LayerNorm(dim)
        self.norm4 = nn.LayerNorm(dim)
        self.cross_attn_image_to_token = Attention(dim, heads, 2, qkv)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe):
        if self.skip_first_layer_pe:
            queries = self.self_attn(queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + self.self_attn(q, q, queries)
        queries = self.norm1(queries)
        q, k = queries + query_pe, keys + key_pe
        queries = self.norm2(queries + self.cross_attn_token_to_image(q, k, keys))
        queries = self.norm3(queries + self.mlp(queries))
        q, k = queries + query_pe, keys + key_pe
        keys = self.norm4(keys + self.cross_attn_image_to_token(k, q, queries))
        return queries, keys


class TwoWayTransformer(nn.Module):
    def __init__(self, depth=2, dim=256, heads=8, mlp_dim=2048, lora=True):
        super().__init__()
        self.layers = nn.ModuleList(TwoWayBlock(dim, heads, mlp_dim, i == 0, lora) for i in range(depth))
        self.final_attn_token_to_image = Attention(dim, heads, 2)
        self.norm_final_attn = nn.LayerNorm(dim)

    def forward(self, image_embedding, image_pe, point_embedding):
        keys = image_embedding.flatten(2).permute(0, 2, 1)
        key_pe = image_pe.flatten(2).permute(0, 2, 1)
        queries = point_embedding
        for layer in self.layers:
            queries, keys = layer(queries, keys, point_embedding, key_pe)
        q, k = queries + point_embedding, keys + key_pe
        queries = self.norm_final_attn(queries + self.final_attn_token_to_image(q, k, keys))
        return queries, keys


class LayerNorm2d(nn.Module):
    def __init__(self, c, eps=1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c)), eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class MaskDecoder(nn.Module):
    def __init__(self, dim=256, num_multimask=3, lora=True, dynamic_multimask_via_stability=True,
                 stability_delta=0.05, stability_thresh=0.98):
        super().__init__()
        self.dim, self.nmt = dim, num_multimask + 1
        self.transformer = TwoWayTransformer(2, dim, 8, 2048, lora)
        self.iou_token = nn.Embedding(1, dim)
        self.mask_tokens = nn.Embedding(self.nmt, dim)
        self.obj_score_token = nn.Embedding(1, dim)
        self.output_upscaling = nn.Sequential(
            nn.ConvTranspose2d(dim, dim // 4, 2, 2), LayerNorm2d(dim // 4), nn.GELU(),
            nn.ConvTranspose2d(dim // 4, dim // 8, 2, 2), nn.GELU())
        self.conv_s0 = LoRAConv1x1(dim, dim // 8) if lora else nn.Conv2d(dim, dim // 8, 1)
        self.conv_s1 = LoRAConv1x1(dim, dim // 4) if lora else nn.Conv2d(dim, dim // 4, 1)
        self.output_hypernetworks_mlps = nn.ModuleList(MLP(dim, dim, dim // 8, 3) for _ in range(self.nmt))
        self.iou_prediction_head = MLP(dim, 256, self.nmt, 3, sigmoid_output=True, lora=(2,) if lora else ())
        self.pred_obj_score_head = MLP(dim, dim, 1, 3)
        self.dynamic, self.delta, self.thresh = dynamic_multimask_via_stability, stability_delta, stability_thresh

    def predict_masks(self, image_embeddings, image_pe, sparse, dense, high_res, repeat_image=False):
        out_tokens = torch.cat((self.obj_score_token.weight, self.iou_token.weight, self.mask_tokens.weight), 0)
        if repeat_image:
            # upstream box / point prompting: ONE image, sparse [P, K, 256]; the image embedding is repeated per prompt
            # and the high-res features broadcast over the prompt axis
            image_embeddings = image_embeddings.repeat_interleave(sparse.shape[0], 0)
        B = image_embeddings.shape[0]
        # B > 1 without repeat_image: B independent images sharing the learned tokens (SURVEY.md 8(a) batch note)
        tokens = torch.cat((out_tokens.unsqueeze(0).expand(B, -1, -1), sparse.expand(B, -1, -1)), 1)
        src = image_embeddings + dense
        pos = image_pe.expand(B, -1, -1, -1)
        b, c, h, w = src.shape
        hs, src = self.transformer(src, pos, tokens)
        iou_tok, mask_toks = hs[:, 1], hs[:, 2:2 + self.nmt]
        src = src.transpose(1, 2).reshape(b, c, h, w)
        dc1, ln1, act1, dc2, act2 = self.output_upscaling
        s0, s1 = high_res
        up = act1(ln1(dc1(src) + s1))
        up = act2(dc2(up) + s0)
        hyper = torch.stack([self.output_hypernetworks_mlps[i](mask_toks[:, i]) for i in range(self.nmt)], 1)
        b, c, h, w = up.shape
        masks = (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)
        return masks, self.iou_prediction_head(iou_tok), self.pred_obj_score_head(hs[:, 0])

    def forward(self, image_embeddings, image_pe, sparse, dense, high_res, multimask_output=False, repeat_image=False):
        masks, iou, obj = self.predict_masks(image_embeddings, image_pe, sparse, dense, high_res, repeat_image)
        if multimask_output:
            return masks[:, 1:], iou[:, 1:], obj
        if self.dynamic and not self.training:
            m, i = self._dynamic(masks, iou)
            return m, i, obj
        return masks[:, 0:1], iou[:, 0:1], obj

    def _dynamic(self, masks, iou):
        multi, multi_iou = masks[:, 1:], iou[:, 1:]
        best = torch.argmax(multi_iou, -1)
        ar = torch.arange(masks.shape[0])
        best_m, best_i = multi[ar, best].unsqueeze(1), multi_iou[ar, best].unsqueeze(1)
        single, single_i = masks[:, 0:1], iou[:, 0:1]
        flat = single.flatten(-2)
        area_i = (flat > self.delta).sum(-1).float()
        area_u = (flat > -self.delta).sum(-1).float()
        stab = torch.where(area_u > 0, area_i / area_u, torch.ones_like(area_u))
        ok = stab >= self.thresh
        return torch.where(ok[..., None, None], single, best_m), torch.where(ok, single_i, best_i)


# ---- refinement head, wrapper, transforms (reference-owned code paths) ---------------------------------------------
class MultiKernelRefinement(nn.Module):
    """src/sam2_infer.py:130-189: parallel k x k convs (1 -> 4 ch, 'same' zero pad) -> exact GELU -> cat -> 1x1."""

    def __init__(self, kernel_sizes=(3, 5, 7, 11), intermediate_channels=4):
        super().__init__()
        self.conv_branches = nn.ModuleList(nn.Conv2d(1, intermediate_channels, k, padding=k // 2) for k in kernel_sizes)
        self.combiner_conv = nn.Conv2d(len(kernel_sizes) * intermediate_channels, 1, 1)

    def forward(self, x):
        return self.combiner_conv(torch.cat([F.gelu(b(x)) for b in self.conv_branches], 1))


HIERA_L = dict(embed_dim=144, num_heads=2, stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43), window_spec=(8, 4, 16, 8))
HIERA_T = dict(embed_dim=96, num_heads=1, stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9), window_spec=(8, 4, 14, 7))
# reference LoRA targets inside the trunk (circuit_analyzer.py:186-191), valid for Hiera-L block numbering
LORA_TRUNK_L = {44: ("attn.qkv", "mlp.layers.0", "proj"), 47: ("attn.qkv", "mlp.layers.0")}


class SAM2Core(nn.Module):
    """The parts of SAM2Base the wrapper touches, with upstream parameter names."""

    def __init__(self, hiera=HIERA_L, lora=True, lora_trunk=None, image_size=1024, dynamic_multimask_via_stability=True):
        super().__init__()
        self.image_size = image_size
        if lora_trunk is None:
            lora_trunk = LORA_TRUNK_L if (lora and hiera is HIERA_L) else {}
        self.image_encoder = nn.Module()
        self.image_encoder.trunk = Hiera(**hiera, lora_blocks=lora_trunk)
        chans = self.image_encoder.trunk.channel_list[::-1]
        self.image_encoder.neck = FpnNeck(chans, 256, (2, 3), lora_convs=(2, 3) if lora else ())
        self.sam_prompt_encoder = PromptEncoder(256, image_size)
        self.sam_mask_decoder = MaskDecoder(256, 3, lora, dynamic_multimask_via_stability)
        self.no_mem_embed = nn.Parameter(torch.zeros(1, 1, 256))           # added to the image embedding by the image predictor


class SAM2ImageWrapper(nn.Module):
    """src/sam2_infer.py:191-275 (forward :220-275)."""

    def __init__(self, core, embedding_r=4, use_refinement=True, kernel_sizes=(3, 5, 7, 11)):
        super().__init__()
        self.sam2_model = core
        fs = core.image_size // 16
        self.dense_embedding1 = nn.Parameter(torch.randn(1, 256, embedding_r))
        self.dense_embedding2 = nn.Parameter(torch.randn(1, embedding_r, fs * fs))
        self.sparse_embedding = nn.Parameter(torch.randn(1, 32, 256))
        self.refinement_layer = MultiKernelRefinement(kernel_sizes, 4) if use_refinement else None

    def encode(self, images):
        m = self.sam2_model
        fpn = m.image_encoder.neck(m.image_encoder.trunk(images))[:-1]          # scalp = 1
        s0 = m.sam_mask_decoder.conv_s0(fpn[0])
        s1 = m.sam_mask_decoder.conv_s1(fpn[1])
        return fpn[2], [s0, s1]

    def forward(self, images, return_intermediates=False):
        m = self.sam2_model
        embed, high_res = self.encode(images)
        fs = embed.shape[-1]
        dense = (self.dense_embedding1 @ self.dense_embedding2).view(1, 256, fs, fs)
        pe = dense_pe(m.sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix, fs, fs)
        low, iou, _ = m.sam_mask_decoder(embed, pe, self.sparse_embedding, dense, high_res, multimask_output=False)
        high = F.interpolate(low, size=(m.image_size, m.image_size), mode="bilinear", align_corners=False)
        if self.refinement_layer is not None:
            high = self.refinement_layer(high)
        if return_intermediates:
            return high, low, iou, dict(embed=embed, s0=high_res[0], s1=high_res[1])
        return high, low, iou


def predict_boxes(wrapper, images, boxes, multimask_output=False):
    """Upstream SAM2ImagePredictor semantics (set_image_batch + _predict with box prompts) on the wrapper's
    (fine-tuned) weights: per image, P boxes (xyxy, in the model's input pixel space) -> P masks.
      image_embed = FPN level 2 + no_mem_embed   (yaml `directly_add_no_mem_embed: true`, models/configs/sam2.1_hiera_l.yaml:95)
      sparse = corners as points labelled 2 / 3 + one padding point;  dense = no_mask_embed
      decoder with repeat_image=True; multimask_output=False -> token 0, or the best multimask when unstable
    images [B,3,R,R], boxes [B,P,4] -> high_res logits [B,P,R,R] (plain bilinear, no refinement head), low_res [B,P,R/4,R/4], iou [B,P]."""
    m = wrapper.sam2_model
    embed, high_res = wrapper.encode(images)
    embed = embed + m.no_mem_embed.view(1, -1, 1, 1)
    fs = embed.shape[-1]
    pe = dense_pe(m.sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix, fs, fs)
    dense = m.sam_prompt_encoder.dense_no_mask(fs)
    his, lows, ious = [], [], []
    for b in range(images.shape[0]):
        sparse = m.sam_prompt_encoder.embed_boxes(boxes[b].float())
        low, iou, _ = m.sam_mask_decoder(embed[b:b + 1], pe, sparse, dense, [h[b:b + 1] for h in high_res],
                                         multimask_output=multimask_output, repeat_image=True)
        his.append(F.interpolate(low, size=(m.image_size, m.image_size), mode="bilinear", align_corners=False)[:, 0])
        lows.append(low[:, 0]); ious.append(iou[:, 0])
    return torch.stack(his), torch.stack(lows), torch.stack(ious)


def predict_prompts(wrapper, images, boxes=None, points=None, labels=None, multimask_output=False):
    """predict_boxes generalised to the predictor's full sparse prompt (SAM2ImagePredictor._predict): per prompt the box
    corners (labels 2 / 3) first, then the clicks (points [B,P,K,2], labels [B,P,K] in {1, 0, -1}), then one padding point."""
    m = wrapper.sam2_model
    embed, high_res = wrapper.encode(images)
    embed = embed + m.no_mem_embed.view(1, -1, 1, 1)
    fs = embed.shape[-1]
    pe = dense_pe(m.sam_prompt_encoder.pe_layer.positional_encoding_gaussian_matrix, fs, fs)
    dense = m.sam_prompt_encoder.dense_no_mask(fs)
    his, lows, ious = [], [], []
    for b in range(images.shape[0]):
        cs, ls = [], []
        if boxes is not None:
            cs.append(boxes[b].float().reshape(-1, 2, 2))
            ls.append(torch.tensor([[2, 3]], dtype=torch.long).expand(boxes.shape[1], 2))
        if points is not None:
            cs.append(points[b].float())
            ls.append(labels[b].long())
        sparse = m.sam_prompt_encoder.embed_points(torch.cat(cs, 1), torch.cat(ls, 1), pad=True)
        low, iou, _ = m.sam_mask_decoder(embed[b:b + 1], pe, sparse, dense, [h[b:b + 1] for h in high_res],
                                         multimask_output=multimask_output, repeat_image=True)
        his.append(F.interpolate(low, size=(m.image_size, m.image_size), mode="bilinear", align_corners=False)[:, 0])
        lows.append(low[:, 0]); ious.append(iou[:, 0])
    return torch.stack(his), torch.stack(lows), torch.stack(ious)


def postprocess_masks(masks, orig_hw):
    """src/sam2_infer.py:88-128 with max_hole_area = max_sprinkle_area = 0 (circuit_analyzer.py:245-250)."""
    return F.interpolate(masks.float(), orig_hw, mode="bilinear", align_corners=False)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def sam2_transform(img_u8, resolution=1024):
    """src/sam2_infer.py:49-51: ToTensor (u8 HWC -> f32 CHW / 255), Resize((R, R)) on a tensor (bilinear,
    antialias=True in current torchvision -> torch's _upsample_bilinear2d_aa), Normalize(ImageNet)."""
    x = torch.from_numpy(img_u8).permute(2, 0, 1).float().div(255)
    x = F.interpolate(x[None], size=(resolution, resolution), mode="bilinear", align_corners=False, antialias=True)[0]
    mean, std = torch.tensor(IMAGENET_MEAN).view(3, 1, 1), torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (x - mean) / std


def randomize_(model, seed=0, std=0.02):
    """Seeded synthetic weights: trunc_normal(0.02) linears/convs, non-zero LoRA B."""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or "norm" in name and name.endswith("weight"):
            p.data.uniform_(0.8, 1.2, generator=g)
        elif p.dim() >= 2:
            p.data.normal_(0, std, generator=g).clamp_(-2 * std, 2 * std)
        else:
            p.data.normal_(0, std, generator=g)
    return model
