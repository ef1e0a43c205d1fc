"""`fmc_conv3x3_halo_bf16` (csrc/conv_halo.hip): the 3x3 convolution with its input halo resident in LDS and GroupNorm + SiLU applied while the
halo is staged -- SURVEY.md section 8 f1, reference sites diffusers ResnetBlock2D (fmc/models/unet_blocks.py:306-317) and Upsample2D (:625).

Checked against a plain PyTorch fp32 restatement on the same bf16-rounded operands (products are exact in fp32, so the bound is the bf16 rounding
of the output: 2^-8 relative to the largest output; 6e-3 asserted), and -- the fused GroupNorm -- against `oracle.diffusers_restated.ResnetBlock2D`."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_inf(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _mk(n, hs, ws, cin, cout, seed=0, c2=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, hs, ws, cin - c2, generator=g).bfloat16()
    x2 = torch.randn(n, hs, ws, c2, generator=g).bfloat16() if c2 else None
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    return x, x2, w, g


def _ref(x, x2, w, bias=None, temb=None, res=None, temb_div=1, upsample=False, coef=None, act=True):
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    if coef is not None:
        z = xin * coef[:, None, None, :, 0] + coef[:, None, None, :, 1]
        xin = (F.silu(z) if act else z).bfloat16().float()             # the operand is rounded to bf16 exactly once, like the stored GroupNorm output
    xin = xin.permute(0, 3, 1, 2)
    if upsample:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    y = F.conv2d(xin, w.float(), None if bias is None else bias.float(), padding=1)
    if temb is not None:
        y = y + temb.float().repeat_interleave(temb_div, 0)[:, :, None, None]
    if res is not None:
        y = y + res.float().permute(0, 3, 1, 2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("n,h,w,cin,cout,c2,ups,extras", [
    (2, 20, 32, 64, 160, 0, False, False),       # one chunk, two pixel tiles per image
    (3, 10, 64, 128, 320, 0, False, True),       # two channel tiles, bias + temb + residual
    (2, 24, 32, 192, 160, 0, False, True),       # a tile hanging over the last image row (rows 20 .. 23 + 6 masked)
    (2, 20, 64, 128, 160, 0, True, True),        # nearest 2x upsample folded into the halo addressing
    (2, 20, 32, 320, 320, 128, False, True),     # two-source input (channel concat read in place), 5 chunks across the seam
    (1, 7, 32, 64, 160, 0, False, False),        # an image lower than one tile
])
def test_conv3x3_halo_matches_fp32_conv(n, h, w, cin, cout, c2, ups, extras):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from synfmc_amd import hip_ops as K
    hs, ws = (h // 2, w // 2) if ups else (h, w)
    assert K.conv3x3_halo_supported(n, h, w, cin, cin - c2, cout, ups)
    x, x2, wt, g = _mk(n, hs, ws, cin, cout, seed=h + cin, c2=c2)
    bias = temb = res = None
    if extras:
        bias = torch.randn(cout, generator=g).bfloat16()
        temb = torch.randn(1, cout, generator=g).bfloat16() if n % 2 else torch.randn(n // 2, cout, generator=g).bfloat16()
        res = torch.randn(n, h, w, cout, generator=g).bfloat16()
    div = n // temb.shape[0] if temb is not None else 1
    want = _ref(x, x2, wt, bias, temb, res, div, ups)
    cu = lambda t: None if t is None else t.cuda()
    got = K.conv3x3_halo(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2))
    torch.cuda.synchronize()
    assert got.shape == want.shape
    assert rel_inf(got, want) < 6e-3


@pytest.mark.parametrize("n,h,w,cin,cout,c2,act", [(2, 20, 32, 320, 320, 0, True), (2, 10, 32, 640, 640, 320, True), (2, 20, 32, 128, 320, 0, False)])
def test_conv3x3_halo_groupnorm_prologue_and_statistics_epilogue(n, h, w, cin, cout, c2, act):
    """GroupNorm(32) + SiLU in the conv's operand path, statistics of the output for the next GroupNorm out of its epilogue: against
    `conv(silu(group_norm(x)))` in fp32 (torch), the coefficients from `fmc_groupnorm_coef` on exact partial sums."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from synfmc_amd import hip_ops as K
    x, x2, wt, g = _mk(n, h, w, cin, cout, seed=cin + cout, c2=c2)
    x = (x.float() * 1.7 + 0.3).bfloat16()
    gamma, beta = torch.randn(cin, generator=g) * 0.3 + 1.0, torch.randn(cin, generator=g) * 0.2
    xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], -1)
    G, eps = 32, 1e-5
    # partial sums as a producer's epilogue would leave them: [n, splits, G, 2] (here: two splits over the pixels)
    xs = xin.reshape(n, 2, h * w // 2, G, cin // G)
    part = torch.stack([xs.sum((2, 4)), (xs * xs).sum((2, 4))], -1).contiguous()
    coef, stats = K.groupnorm_coef(part.cuda(), gamma.cuda(), beta.cuda(), h * w, cin, G, eps, want_stats=True)
    gn = F.group_norm(xin.permute(0, 3, 1, 2), G, gamma, beta, eps)
    mean = xin.reshape(n, h * w, G, cin // G).mean((1, 3))
    assert rel_inf(stats[..., 0], mean) < 1e-5
    z_ref = gn.permute(0, 2, 3, 1)
    z_dev = xin * coef.cpu()[:, None, None, :, 0] + coef.cpu()[:, None, None, :, 1]
    assert rel_inf(z_dev, z_ref) < 1e-5
    want = _ref(x, x2, wt, coef=coef.cpu(), act=act)
    cu = lambda t: None if t is None else t.cuda()
    got, parts = K.conv3x3_halo(cu(x), wt.cuda(), x2_nhwc=cu(x2), gn_coef=coef, gn_act=act, emit_gn=True)
    torch.cuda.synchronize()
    assert rel_inf(got, want) < 8e-3                                       # (+ the hardware exp2 / rcp of the SiLU in front of the bf16 rounding)
    # statistics epilogue: sums of the ROUNDED outputs per (image, group), summed over the pixel tiles
    o = got.float().cpu().reshape(n, h * w, 32, cout // 32)
    s_ref = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)
    assert parts.shape == (n, (h + 9) // 10 * (w // 32), 32, 2)
    assert rel_inf(parts.sum(1), s_ref) < 1e-4


def test_resnet_block_fused_groupnorm_conv_matches_the_oracle_block():
    """diffusers' ResnetBlock2D (the oracle's restatement) against the product block with GroupNorm + SiLU fused into both convolutions."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import diffusers_restated as OD
    from synfmc_amd.models import layers as L
    from synfmc_amd import hip_ops as K
    torch.manual_seed(4)
    cin, cout, n, h, w = 320, 320, 4, 20, 32
    ref = OD.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=1280, groups=32, eps=1e-5)
    with torch.no_grad():
        for p in ref.parameters():
            p.normal_(0, p[0].numel() ** -0.5) if p.ndim >= 2 else p.normal_(0, 0.2)
        ref.norm1.weight.add_(1.0); ref.norm2.weight.add_(1.0)
    blk = L.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=1280, groups=32, eps=1e-5)
    blk.load_state_dict(ref.state_dict(), strict=True)
    blk = blk.to("cuda", torch.bfloat16).eval().requires_grad_(False)
    ref = ref.bfloat16().float()
    x = torch.randn(n, cin, h, w).bfloat16()
    temb = torch.randn(n, 1280).bfloat16()
    min_tiles, K.CONV_HALO_MIN_TILES = K.CONV_HALO_MIN_TILES, 1                 # (a 4-image test input: 16 workgroups; the step's convolutions launch 256 - 1024)
    with torch.no_grad():
        want = ref(x.float(), temb.float())
        before = dict(K.conv_halo_calls)
        K.CONV_GN_FUSED = True                                                    # (opt-in: measured slower than the separate apply pass)
        try:
            got = blk(x.cuda().contiguous(memory_format=torch.channels_last), temb.cuda())
        finally:
            K.CONV_GN_FUSED = False
        assert K.conv_halo_calls["gn_fused"] - before["gn_fused"] == 2           # both convolutions took the fused path
        assert got._fmc_gn[0].shape == (n, 2, 32, 2)                              # ... and left the statistics for the next GroupNorm
        halo0 = K.conv_halo_calls["conv"]
        plain = blk(x.cuda().contiguous(memory_format=torch.channels_last), temb.cuda())     # default: halo conv + separate GroupNorm apply
        assert K.conv_halo_calls["conv"] - halo0 == 2 and K.conv_halo_calls["gn_fused"] - before["gn_fused"] == 2
    K.CONV_HALO_MIN_TILES = min_tiles
    e_f, e_p = rel_inf(got, want), rel_inf(plain, want)
    assert e_f < 2e-2 and e_f < 2.0 * e_p + 2e-3, (e_f, e_p)


@pytest.mark.parametrize("n,h,w,cin,cout,c2,ups,extras", [
    (4, 10, 16, 64, 80, 0, False, False),        # two images per tile, one chunk
    (5, 10, 16, 192, 160, 0, False, True),       # an odd image count (the last tile holds one image), bias + temb (per image of a tile) + residual
    (6, 10, 16, 256, 160, 128, False, True),     # two-source input
    (4, 10, 16, 128, 80, 0, True, True),         # nearest 2x upsample (5x8 -> 10x16)
    (16, 5, 8, 128, 160, 0, False, True),        # 5 x 8 images: eight per tile, fragments spanning two image rows
    (11, 5, 8, 64, 80, 0, False, False),         # ... with a last tile of three images
    (3, 16, 16, 64, 80, 0, False, True),         # 16 x 16 images (the 32x512x512 configuration): row blocks of 10 + 6 rows
    (2, 20, 32, 128, 160, 0, False, True),       # 32 pixels wide: two 10 x 16 row blocks side by side
    (2, 32, 48, 64, 80, 0, False, True),         # the reference's training clip (256x384 -> 32 x 48 latents): 4 x 3 row blocks of 10 x 16, rows 30-31 in a short block
    (3, 16, 24, 128, 160, 64, False, True),      # ... its 16 x 24 level: 5 x 8 row blocks (4 x 3 per image, the last row block one row high), two-source
])
def test_conv3x3_halo4_matches_fp32_conv(n, h, w, cin, cout, c2, ups, extras):
    """`fmc_conv3x3_halo4_bf16` (csrc/conv_halo4.hip), the halo-resident convolution of the small feature maps, against the fp32 convolution of the
    same bf16-rounded operands; the statistics epilogue against the sums of the rounded outputs."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from synfmc_amd import hip_ops as K
    hs, ws = (h // 2, w // 2) if ups else (h, w)
    assert K.conv3x3_halo4_supported(n, h, w, cin, cin - c2, cout, ups)
    wide_ok = K.conv3x3_halo4_supported(n, h, w, cin, cin - c2, cout, ups, wide=True)
    x, x2, wt, g = _mk(n, hs, ws, cin, cout, seed=h + cin + n, c2=c2)
    bias = temb = res = None
    div = 1
    if extras:
        bias = torch.randn(cout, generator=g).bfloat16()
        div = 2 if n % 2 == 0 else 1
        temb = torch.randn(n // div, cout, generator=g).bfloat16()
        res = torch.randn(n, h, w, cout, generator=g).bfloat16()
    want = _ref(x, x2, wt, bias, temb, res, div, ups)
    cu = lambda t: None if t is None else t.cuda()
    emit = cout % 64 == 0 and 80 % (cout // 32) == 0
    got = K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2), emit_gn=emit, split_k=1)
    torch.cuda.synchronize()
    if emit:
        got, parts = got
        o = got.float().cpu().reshape(n, h * w, 32, cout // 32)
        s_ref = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)
        assert parts.shape[0] == n and parts.shape[2:] == (32, 2)
        assert rel_inf(parts.sum(1), s_ref) < 1e-4
    assert got.shape == want.shape
    assert rel_inf(got, want) < 6e-3
    again = K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2), split_k=1)
    assert torch.equal(again, got)                                           # deterministic
    if cin >= 128:                                                           # split-K: the chunks dealt to 2 (and cin / 64) workgroups per tile, fp32 partials + finishing pass
        for sk in (2, cin // 64):
            split = K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2), split_k=sk)
            assert rel_inf(split, want) < 6e-3
            assert torch.equal(split, K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2), split_k=sk))
        auto = K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2))     # (the front-end's own choice of split)
        assert rel_inf(auto, want) < 6e-3
    if wide_ok:                                                              # the 8-wave form (160 channels per tile): same results as the 4-wave form, bit for bit
        wide = K.conv3x3_halo4(cu(x), wt.cuda(), cu(bias), cu(temb), cu(res), temb_div=div, upsample=ups, x2_nhwc=cu(x2), split_k=1, wide=True, emit_gn=emit)
        if emit:
            wide, wparts = wide
            assert rel_inf(wparts.sum(1), parts.sum(1)) < 1e-5
        assert torch.equal(wide, got)
