"""Split-fp16 GEMM (csrc/gemm_planes.hip, through the C ABI) against fp64: the result must be as close to the exact
product as an fp32 GEMM is.  Error measure: |C - C64| / (|A| @ |B|) (error relative to the magnitude of the terms
of each dot product), the same measure as scripts/mb/split_mfma_accuracy.hip."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'GPU test on a box without GPU'
    return torch.device('cuda:0')


def _err(c, a, b):
    ref = a.double() @ b.double()
    mag = a.double().abs() @ b.double().abs()
    return float(((c.double() - ref).abs() / mag.clamp_min(1e-300)).max())


SHAPES = [(1, 1, 1), (5, 3, 2), (128, 128, 32), (130, 257, 33), (257, 130, 514), (300, 514, 1200), (1012, 1200, 257),
          (2024, 4800, 1200), (8096, 1200, 600)]


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_all_layouts_vs_fp64(M, N, K, ta, tb):
    from padertorch_amd.ops import gemm
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    a = (torch.rand(M, K, generator=g) * 2 - 1).to(dev)
    b = (0.05 * torch.randn(K, N, generator=g)).to(dev)
    xa = a.t().contiguous().t() if ta else a          # [M, K] view of [K][M] storage
    xb = b.t().contiguous().t() if tb else b          # [K, N] view of [N][K] storage
    c = gemm.mm(xa, xb)
    e = _err(c, a, b)
    e32 = _err(a @ b, a, b)
    assert e < 4e-7, (e, e32)


def test_wide_dynamic_range_bias_accumulate_views():
    """Gradient-like operand (values over many octaves, tiny overall scale), bias, accumulation into a strided output,
    operands that are column blocks of wider matrices, split-K."""
    from padertorch_amd.ops import gemm
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(3)
    M, N, K = 700, 260, 2400
    wide = (1e-6 * (torch.rand(M, 2 * K, generator=g) * 2 - 1) * torch.exp(8 * (torch.rand(M, 2 * K, generator=g) * 2 - 1))).to(dev)
    a = wide[:, K:]                                   # column block, row stride 2K
    w = (0.05 * torch.randn(K, N, generator=g)).to(dev)
    bias = torch.randn(N, generator=g).to(dev) * 1e-6
    c = gemm.mm(a, w, bias=bias)
    ref = a.double() @ w.double() + bias.double()
    mag = a.double().abs() @ w.double().abs() + bias.double().abs()
    e32 = float((((a @ w + bias).double() - ref).abs() / mag).max())
    assert float(((c.double() - ref).abs() / mag).max()) < max(4e-7, 2 * e32), e32
    # weight-gradient form: out[N, M] += w^T-like reduction over the rows, into a view of a flat buffer
    flat = torch.zeros(N * M + 5, device=dev)
    out = flat[5:].view(N, M)
    out.fill_(1e-7)
    x = (torch.rand(K, N, generator=g) * 2 - 1).to(dev)
    dg = wide[:K, :M] if wide.shape[0] >= K else None
    assert dg is None
    dg = (1e-5 * torch.randn(K, M, generator=g)).to(dev)
    for split in (1, 4):
        out.fill_(1e-7)
        gemm.mm(x.t(), dg, out=out, accumulate=True, split_k=split)
        ref = x.double().t() @ dg.double() + 1e-7
        mag = x.double().abs().t() @ dg.double().abs() + 1e-7
        assert float(((out.double() - ref).abs() / mag).max()) < 4e-7, split
    assert float(flat[:5].abs().max()) == 0.


def test_one_product_mode_is_reduced_precision(monkeypatch):
    """``ops.gemm.PRODUCTS = 1`` (BASELINE configs[1]'s "bf16" run): only the hi planes are multiplied - plain 16-bit operands
    (here fp16 halves of the scaled values: 11 significant bits), fp32 accumulation."""
    from padertorch_amd.ops import gemm
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(5)
    a = (torch.rand(300, 640, generator=g) * 2 - 1).to(dev)
    b = (0.05 * torch.randn(640, 200, generator=g)).to(dev)
    full = gemm.mm(a, b)
    monkeypatch.setattr(gemm, 'PRODUCTS', 1)
    c = gemm.mm(a, b)
    assert 1e-6 < _err(c, a, b) < 2e-3 and _err(full, a, b) < 4e-7         # 11-bit operands vs 22-bit ones
    # the bf16 planes (the gate gradients' route): hi halves = bf16-rounded values
    pa = torch.ops.ptmi.pack_planes_bf16(a, False)
    pw = torch.ops.ptmi.pack_planes_bf16(b, True)
    y = torch.empty(300, 200, device=dev)
    torch.ops.ptmi.gemm_planes_bf16_(y, pa, 0, pw, None, 300, 200, 640, False, 1)
    ref16 = a.bfloat16().double() @ b.bfloat16().double()
    assert float((y.double() - ref16).abs().max()) < 1e-4


def test_padded_row_stride_views_take_the_aligned_path():
    """The layer-0 shapes (F = 257): the projection multiplies zero-padded operands (K = 260), the weight gradient reads the
    [rows, 257] view of the padded input (row stride 260) - also when that view ends exactly at its last valid element."""
    from padertorch_amd.ops import gemm
    dev = _dev()
    g = torch.Generator(device='cpu').manual_seed(11)
    rows, I, G = 1012, 257, 640
    x = (torch.rand(rows, I, generator=g) * 3).to(dev)
    w = (0.05 * torch.randn(G, I, generator=g)).to(dev)
    dg = (1e-3 * torch.randn(rows, G, generator=g)).to(dev)
    xp = torch.nn.functional.pad(x, (0, 3))
    out = gemm.mm(xp, torch.nn.functional.pad(w, (0, 3)).t())
    assert _err(out, x, w.t()) < 4e-7
    xv = xp[:, :I]
    assert xv.stride(0) == 260
    dw = gemm.mm(dg.t(), xv, split_k=1)
    assert _err(dw, dg.t(), x) < 4e-7
    # a view whose storage ends with the last valid element (no padding behind the last row)
    flat = torch.zeros((rows - 1) * 260 + I, device=dev)
    tight = flat.as_strided((rows, I), (260, 1))
    tight.copy_(x)
    dw2 = gemm.mm(dg.t(), tight, split_k=1)
    assert _err(dw2, dg.t(), x) < 4e-7


@pytest.mark.parametrize('M,N,K,split,acc', [(2400, 1200, 8096, None, True), (2400, 600, 8096, 4, False), (2400, 257, 8096, None, True),
                                             (100, 36, 70, 1, False), (129, 130, 31, 2, True), (16, 4, 3000, 3, False)])
def test_planes_gemm_vs_fp64(M, N, K, split, acc):
    """pack_planes_t + gemm_planes_ (the weight-gradient form: both operands reduce over their outer axis) against fp64,
    with the bound of the in-register split GEMM; strided sources, odd sizes, split K, accumulation into a strided C."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(M + N + K)
    big_a = torch.randn(K, M + 5, device='cuda') * 3.0                  # sources are column blocks of wider matrices
    big_b = torch.randn(K, N + 3, device='cuda') * 0.02
    a, b = big_a[:, 2:2 + M], big_b[:, 1:1 + N]
    cbuf = torch.randn(M, N + 2, device='cuda')
    c = cbuf[:, :N]
    want = a.double().t() @ b.double() + (c.double() if acc else 0)
    mag = a.double().abs().t() @ b.double().abs()
    G.mm_planes_(c, G.pack_t(a), G.pack_t(b), M, N, K, accumulate=acc, split_k=split)
    err = float(((c.double() - want).abs() / mag).max())
    assert err < 4e-7, err
    again = cbuf.clone()[:, :N]
    # bitwise reproducible (slabs summed in order)
    c2 = (torch.zeros(M, N, device='cuda') if not acc else None)
    if c2 is not None:
        G.mm_planes_(c2, G.pack_t(a), G.pack_t(b), M, N, K, split_k=split)
        assert torch.equal(c2, again)


def test_planes_pack_layout_and_unit_range():
    """The plane layout is what csrc/gemm_planes.hip documents; UNIT_RANGE packs without a scale."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(0)
    x = (torch.rand(70, 20, device='cuda') * 2 - 1)
    planes, word = G.pack_t(x, G.UNIT_RANGE)
    assert word is None
    p = planes.view(2, 3, 2, 4, 16, 8).cpu().float()          # [row tile][k block][plane][k group][row][8]
    xs = x.cpu()
    for c in (0, 7, 19):
        for k in (0, 31, 32, 69):
            hi = p[c // 16, k // 32, 0, (k % 32) // 8, c % 16, k % 8]
            lo = p[c // 16, k // 32, 1, (k % 32) // 8, c % 16, k % 8]
            assert float(hi) == float(xs[k, c].half()) and abs(float(hi + lo) - float(xs[k, c])) < 2e-7
    assert float(p[1, :, :, :, 4:, :].abs().sum()) == 0 and float(p[:, 2, :, 0, :, 6:].abs().sum()) == 0      # past the matrix


def test_mm_with_a_single_column_or_row_operand():
    """``ops.gemm.mm`` with K = 1 / N = 1 / M = 1: a size-1 axis keeps whatever stride it had (``.contiguous()`` does not touch it), the
    pack passes must not take it for the inner stride (found by scripts/fuzz_lstm.py: an LSTM with ONE input feature)."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(2)
    x = torch.randn(8192, 3, device='cuda')[:, 1:2]                  # [8192, 1], strides (3, 1)
    xt = torch.randn(1, 8192, device='cuda').t()                     # [8192, 1], strides (1, 8192)
    w = torch.randn(1, 40, device='cuda')
    for a in (x, xt):
        assert float((G.mm(a, w).double() - a.double() @ w.double()).abs().max()) < 1e-5
    v = torch.randn(300, 1, device='cuda')
    m = torch.randn(70, 300, device='cuda')
    assert float((G.mm(m, v).double() - m.double() @ v.double()).abs().max()) < 1e-4
    r = torch.randn(1, 300, device='cuda')
    assert float((G.mm(r, m.t().contiguous()).double() - r.double() @ m.double().t()).abs().max()) < 1e-4
    assert float((G.mm(r.expand(1, 300), m.t()).double() - r.double() @ m.double().t()).abs().max()) < 1e-4


def test_absmax_words_and_running_maximum():
    """``ops.gemm.absmax``: the float bits of max |x| in a word of a buffer zeroed once (no zeroing launch per call; every call its
    own word, per stream), for either unit stride; ``ptmi_absmax_accumulate`` keeps a running maximum; ``ptmi_absmax`` zeroes."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(3)
    xs = [torch.randn(300, 70, device='cuda') * s for s in (1., 5., 0.2)]
    words = [G.absmax(x) for x in xs] + [G.absmax(xs[1][:, 3:50]), G.absmax(xs[1].t())]
    assert len({w.data_ptr() for w in words}) == len(words)
    for w, x in zip(words, xs + [xs[1][:, 3:50], xs[1].t()]):
        assert float(w.view(torch.float32)) == float(x.abs().max())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.default_stream())
        w2 = G.absmax(xs[2])
    side.synchronize()
    assert float(w2.view(torch.float32)) == float(xs[2].abs().max())
    lib = _lib.load()
    run = torch.zeros(1, dtype=torch.int32, device='cuda')
    for x in xs:
        _lib.check(lib.ptmi_absmax_accumulate(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), run.data_ptr(), _lib.stream(x.device)))
    assert float(run.view(torch.float32)) == max(float(x.abs().max()) for x in xs)
    _lib.check(lib.ptmi_absmax(xs[2].data_ptr(), 300, 70, 70, run.data_ptr(), _lib.stream(run.device)))       # zeroes first
    assert float(run.view(torch.float32)) == float(xs[2].abs().max())


@pytest.mark.parametrize('K,C,ld', [(8096, 2400, 4800), (1000, 514, 516), (70, 18, 20), (33, 257, 260), (5000, 1200, 1200)])
def test_planes_pack_t_aligned_path_equals_scalar_path(K, C, ld):
    """The float4 form of the transposing pack (16-byte aligned rows) writes the same planes as the scalar form (the same values
    from a source shifted by one float), fp16 and bf16, ragged k and column tails."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(K + C)
    wide = torch.randn(K, ld, device='cuda')
    src = wide[:, :C]
    shifted = torch.zeros(K * ld + 1, device='cuda')[1:].view(K, ld)
    shifted.copy_(wide)
    assert src.data_ptr() % 16 == 0 and shifted.data_ptr() % 16 != 0
    amax = G.absmax(src)
    a, b = G.pack_t(src, amax), G.pack_t(shifted[:, :C], amax)
    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))
    a16 = torch.ops.ptmi.pack_planes_bf16(src, True)
    b16 = torch.ops.ptmi.pack_planes_bf16(shifted[:, :C], True)
    assert torch.equal(a16.view(torch.int16), b16.view(torch.int16))


@pytest.mark.parametrize('n,ndir,H,cols', [(4800, 2, 600, 608), (1200, 2, 600, 608), (70, 2, 20, 32), (33, 1, 40, 64)])
def test_planes_pack_into_wider_operand_equals_padded_copy(n, ndir, H, cols):
    """``pack_planes_into_`` (a direction's column / row block into its k blocks of a wider operand) against the pack of the
    zero-padded fp32 copy: the forward form (``pack_n_direction_blocks``) and the bf16 transposed form of the LSTM input
    gradient's weights (``stacked_planes_t_bf16``), bit for bit."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(n + H)
    w = torch.randn(n, ndir * H, device='cuda') * 0.1
    amax = G.absmax(w)
    direct = G.pack_n_direction_blocks(w, ndir, H, cols, amax)
    padded = G.pack_n(G.pad_direction_blocks(w, ndir, H, cols), amax)[0]
    assert torch.equal(direct.view(torch.int16), padded.view(torch.int16))
    # transposed bf16 form: w2 [ndir * g, I], every direction's g rows padded to `cols2` k columns
    g, cols2 = 4 * H, (4 * H + 63) // 64 * 64 + 64
    w2 = torch.randn(ndir * g, n, device='cuda') * 0.1
    direct2 = G.stacked_planes_t_bf16(w2, ndir, cols2)
    wp = w2.new_zeros((ndir, cols2, n))
    wp[:, :g] = w2.view(ndir, g, n)
    padded2 = torch.ops.ptmi.pack_planes_bf16(wp.view(ndir * cols2, n), True)
    assert torch.equal(direct2.view(torch.int16), padded2.view(torch.int16))
    with pytest.raises(RuntimeError):       # k blocks past the operand
        torch.ops.ptmi.pack_planes_into_(direct, w[:, :H], amax, False, ndir * cols // 32, ndir * cols // 32, cols // 32)


@pytest.mark.parametrize('M,N,K,split', [(8096, 4800, 1200, None), (8096, 514, 1200, None), (300, 70, 257, 1), (17, 5, 33, 2)])
def test_planes_forward_form_with_bias(M, N, K, split):
    """x W^T + b with both operands packed from k-contiguous sources (pack_planes_n), odd K / N, strided x."""
    from padertorch_amd.ops import gemm as G
    torch.manual_seed(M + N)
    xb = torch.randn(M, K + 3, device='cuda') * 2.0
    x = xb[:, :K]
    w = torch.randn(N, K, device='cuda') * 0.05
    b = torch.randn(N, device='cuda')
    want = x.double() @ w.double().t() + b.double()
    mag = x.double().abs() @ w.double().abs().t() + b.double().abs()
    y = torch.empty(M, N, device='cuda')
    G.mm_planes_(y, G.pack_n(x), G.pack_n(w), M, N, K, split_k=split, bias=b)
    assert float(((y.double() - want).abs() / mag).max()) < 4e-7


@pytest.mark.parametrize('M,N,K,split', [(8096, 1200, 4800, None), (300, 70, 257, 1), (17, 5, 33, 2)])
def test_bf16_planes_gemm_vs_fp64(M, N, K, split):
    """The bf16 flavour (no operand scale; what the LSTM input gradient runs on): A from a k-contiguous source, B = W^T from
    a source whose reduction axis is the outer one, wide dynamic range, against fp64.  bf16 halves carry 16 mantissa bits:
    the bound is that of the backward recurrence's own products (csrc/lstm_split.hip)."""
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device='cuda') * torch.logspace(-6, 2, K, device='cuda')
    w = torch.randn(K, N, device='cuda') * 0.05
    want = a.double() @ w.double()
    mag = a.double().abs() @ w.double().abs()
    pa = torch.ops.ptmi.pack_planes_bf16(a, False)
    pw = torch.ops.ptmi.pack_planes_bf16(w, True)
    from padertorch_amd.ops import gemm as G
    y = torch.empty(M, N, device='cuda')
    sk = G.auto_split_k(M, N, K) if split is None else split
    torch.ops.ptmi.gemm_planes_bf16_(y, pa, 0, pw, None, M, N, K, False, sk)
    err = float(((y.double() - want).abs() / mag).max())
    assert err < 1.2e-5, err          # 2^-17 per product at worst (two 8-bit halves); sums of many terms average far below
    y2 = torch.empty(M, N, device='cuda')
    torch.ops.ptmi.gemm_planes_bf16_(y2, pa, 0, pw, None, M, N, K, False, sk)
    assert torch.equal(y, y2)


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('M,N,K,acc,bias', [(1000, 700, 257, False, True), (530, 1282, 96, True, False), (257, 321, 1200, True, True),
                                            (16, 4, 64, False, False)])
def test_planes_big_tiles_vs_fp64(tile, M, N, K, acc, bias):
    """Every workgroup tile of the persistent big-tile kernel (PTMI_GEMM_TILE through _lib.select_gemm_tile; 5 = the 128 x 128 kernel): rows /
    columns that end inside a tile, inside an MFMA tile and inside a 4-column store group, several tiles per workgroup
    (more tiles than CUs at 128-wide tiles is not reachable at test sizes - the flat tile loop is exercised through tiles > grid / 8
    per XCD range), bias, accumulation into a strided C; bit-identical between two calls."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import gemm as G
    lib = _lib.load()
    torch.manual_seed(M + N + K + tile)
    x = torch.randn(M, K, device='cuda') * 2.0
    w = torch.randn(N, K, device='cuda') * 0.05
    b = torch.randn(N, device='cuda') if bias else None
    cbuf = torch.randn(M, N + 4, device='cuda')
    c0 = cbuf.clone()
    c = cbuf[:, :N]
    want = x.double() @ w.double().t() + (b.double() if bias else 0) + (c.double() if acc else 0)
    mag = x.double().abs() @ w.double().abs().t() + (b.double().abs() if bias else 0) + (c.double().abs() if acc else 0)
    pa, pb = G.pack_n(x), G.pack_n(w)
    _lib.select_gemm_tile(tile)
    try:
        G.mm_planes_(c, pa, pb, M, N, K, accumulate=acc, split_k=1, bias=b)
        torch.cuda.synchronize()
        assert float(((c.double() - want).abs() / mag).max()) < 4e-7
        assert torch.equal(cbuf[:, N:], c0[:, N:])          # nothing written past the matrix
        again = c0.clone()
        G.mm_planes_(again[:, :N], pa, pb, M, N, K, accumulate=acc, split_k=1, bias=b)
        assert torch.equal(again, cbuf)
    finally:
        _lib.select_gemm_tile(-1)


@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, 5, -1])
@pytest.mark.parametrize('M,N,K,split,acc,bias,bf16', [
    (2400, 600, 2080, 5, True, False, True), (530, 257, 1000, 3, False, True, False), (300, 70, 4111, 12, True, True, True),
    (2400, 1200, 8096, 12, True, False, True), (129, 321, 96, 2, False, False, False)])
def test_planes_big_tiles_with_split_k_vs_fp64(tile, M, N, K, split, acc, bias, bf16):
    """Round 4: split K on the persistent big-tile kernel - work item = (k range, tile), partial products into slabs, summed in slab
    order by the reduction pass (weight-gradient shapes: dW = dgates^T [x | h_prev]).  Every tile pinned (split K as asked for) and the
    cost model's own choice of tile and number of ranges (-1): ragged last k range, more items than CUs per XCD range, rows / columns
    ending inside tiles, bias, accumulation into a strided C, both plane flavours; bit-identical between two calls."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import gemm as G
    lib = _lib.load()
    torch.manual_seed(M + N + K + tile)
    a = torch.randn(K, M, device='cuda') * 0.3            # both operands reduce over their outer axis (the weight-gradient form)
    x = torch.randn(K, N, device='cuda')
    b = torch.randn(N, device='cuda') if bias else None
    cbuf = torch.randn(M, N + 3, device='cuda')
    c0 = cbuf.clone()
    c = cbuf[:, :N]
    want = a.double().t() @ x.double() + (b.double() if bias else 0) + (c.double() if acc else 0)
    mag = a.double().abs().t() @ x.double().abs() + (b.double().abs() if bias else 0) + (c.double().abs() if acc else 0)

    def run(out):
        if bf16:
            torch.ops.ptmi.gemm_planes_bf16_(out, torch.ops.ptmi.pack_planes_bf16(a, True), 0, torch.ops.ptmi.pack_planes_bf16(x, True), b,
                                             M, N, K, acc, split)
        else:
            G.mm_planes_(out, G.pack_t(a), G.pack_t(x), M, N, K, accumulate=acc, split_k=split, bias=b)
    _lib.select_gemm_tile(tile)
    try:
        run(c)
        torch.cuda.synchronize()
        assert float(((c.double() - want).abs() / mag).max()) < (4e-6 if bf16 else 4e-7)
        assert torch.equal(cbuf[:, N:], c0[:, N:])
        again = c0.clone()
        run(again[:, :N])
        assert torch.equal(again, cbuf)
    finally:
        _lib.select_gemm_tile(-1)


@pytest.mark.parametrize('tile', [-1, 0, 3, 5])
@pytest.mark.parametrize('M,split_at,N,K,split,acc', [(4800, 2400, 257, 2080, 8, True), (4800, 2400, 1200, 1000, -1, True),
                                                      (80, 32, 70, 333, 3, False), (272, 256, 321, 96, 1, True)])
def test_planes_gemm_with_a_two_part_output_vs_fp64(tile, M, split_at, N, K, split, acc):
    """ptmi_gemm_planes_bf16_two: rows < split_at of the product land in one buffer, the rest in another (both directions' dW_ih of a
    BLSTM layer in one launch) - every route (cost model, pinned big tiles with and without split K, the 128 x 128 kernel incl. its
    co-resident form), accumulation, strided outputs; equal to two separate calls bit for bit."""
    from padertorch_amd import _lib
    lib = _lib.load()
    torch.manual_seed(M + N + K + tile)
    a = torch.randn(K, M, device='cuda') * 0.3
    x = torch.randn(K, N, device='cuda')
    pa, px = torch.ops.ptmi.pack_planes_bf16(a, True), torch.ops.ptmi.pack_planes_bf16(x, True)
    buf = torch.randn(M + 4, (N + 7) // 4 * 4, device='cuda')          # (both parts in one alignment class: row stride and gap multiples of 16 B)
    c1, c2 = buf[:split_at, :N], buf[split_at + 4:, :N]
    ref = buf.clone()
    want = a.double().t() @ x.double()
    mag = a.double().abs().t() @ x.double().abs()
    before = torch.cat([c1, c2]).double() if acc else 0
    _lib.select_gemm_tile(tile)
    try:
        torch.ops.ptmi.gemm_planes_bf16_two_(c1, c2, pa, 0, px, M, N, K, acc, split)
        torch.cuda.synchronize()
        got = torch.cat([c1, c2]).double()
        assert float(((got - want - before).abs() / (mag + (before.abs() if acc else 0))).max()) < 8e-6       # bf16 (hi, lo) operands: 2^-17 per product
        assert torch.equal(buf[split_at:split_at + 4], ref[split_at:split_at + 4]) and torch.equal(buf[:, N:], ref[:, N:])
        # two separate calls on the two halves of the operand give the same bits
        r1, r2 = ref[:split_at, :N], ref[split_at + 4:, :N]
        half = int(lib.ptmi_planes_elems(split_at, K)) * 2
        if split_at % 16 == 0 and tile in (-1, 5) and split == 1:
            torch.ops.ptmi.gemm_planes_bf16_(r1, pa, 0, px, None, split_at, N, K, acc, split)
            torch.ops.ptmi.gemm_planes_bf16_(r2, pa, half, px, None, M - split_at, N, K, acc, split)
            assert float((torch.cat([r1, r2]) - torch.cat([c1, c2])).abs().max()) <= 1e-6 * float(torch.cat([c1, c2]).abs().max())
    finally:
        _lib.select_gemm_tile(-1)


def test_planes_big_tile_many_tiles_per_workgroup():
    """More tiles than CUs: every workgroup of the persistent kernel walks several tiles (the next tile's first stage is requested
    during the last k step of the current one); 128 x 256 tiles at 4100 x 4100 = 33 x 17 = 561 tiles."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import gemm as G
    lib = _lib.load()
    torch.manual_seed(5)
    M = N = 4100
    K = 160
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') * 0.1
    want = x.double() @ w.double().t()
    mag = x.double().abs() @ w.double().abs().t()
    pa, pb = G.pack_n(x), G.pack_n(w)
    for tile in (4, 0):
        _lib.select_gemm_tile(tile)
        try:
            y = torch.full((M, N), float('nan'), device='cuda')
            G.mm_planes_(y, pa, pb, M, N, K, split_k=1)
            assert float(((y.double() - want).abs() / mag).max()) < 4e-7
        finally:
            _lib.select_gemm_tile(-1)


@pytest.mark.parametrize('tile', [-1, 0, 3, 5])
@pytest.mark.parametrize('M,I,O', [(8096, 1200, 1200), (8096, 1200, 514), (300, 70, 257), (17, 33, 5), (1000, 64, 1282)])
def test_linear_with_the_relu_in_the_epilogue(tile, M, I, O):
    """``ops.linear.linear(..., activation='relu')``: values, the maximum it leaves behind, and all gradients against the unfused form
    (bit for bit: the same products, the activation applied to the same fp32 value) and fp64; every tile, split K included."""
    import ctypes
    from padertorch_amd import _lib
    from padertorch_amd.ops import linear as L
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(M + I + O)
    lin = torch.nn.Linear(I, O).to(dev)
    x = (torch.randn(M, I, device=dev) * 3).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    gy = torch.randn(M, O, device=dev)
    assert _lib.select_gemm_tile(tile) == 0
    try:
        y = L.linear(lin, x, activation='relu')
        rec = getattr(y, L.AMAX_ATTR)
        assert rec[0] == y._version
        amax = rec[1].view(torch.float32)
        assert float(amax) == float(y.max())
        (y * gy).sum().backward()
        gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
        lin.weight.grad = lin.bias.grad = None
        y2 = torch.relu(L.linear(lin, x2))
        (y2 * gy).sum().backward()
    finally:
        _lib.select_gemm_tile(-1)
    assert torch.equal(y, y2)
    assert torch.equal(x.grad, x2.grad)
    assert torch.equal(gw, lin.weight.grad) and torch.equal(gb, lin.bias.grad)
    ref = torch.relu(x.detach().double() @ lin.weight.detach().double().t() + lin.bias.detach().double())
    mag = x.detach().double().abs() @ lin.weight.detach().double().abs().t() + lin.bias.detach().double().abs()
    assert float(((y.double() - ref).abs() / mag).max()) < 4e-7
    # a second fused layer takes the first one's maximum as its operand scale: same result as measuring it
    lin_b = torch.nn.Linear(O, 33).to(dev)
    z = L.linear(lin_b, y.detach().clone() if False else y)
    z2 = L.linear(lin_b, y2.detach())
    assert torch.equal(z.detach(), z2)


@pytest.mark.parametrize('tile', [-1, 0, 5])
@pytest.mark.parametrize('M,I,O', [(8096, 1200, 1200), (300, 70, 257), (17, 33, 5)])
def test_fused_relu_propagates_nan_like_torch_relu(tile, M, I, O):
    """A NaN in the input (or the weights) of a fused Linear + ReLU comes out as NaN in exactly the entries where ``torch.relu(module(x))``
    has one - ``fmaxf(NaN, 0)`` would have turned it into 0, zero masks and a finite loss the Trainer's non-finite check never sees
    (reference check: ``padertorch/train/trainer.py:622-636``) -, in every tile's epilogue and in the slab reduction; the
    backward mask lets the gradient through where the output is NaN, like ``threshold_backward``."""
    from padertorch_amd import _lib
    from padertorch_amd.ops import linear as L
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(M + I + O)
    lin = torch.nn.Linear(I, O).to(dev)
    x = torch.randn(M, I, device=dev)
    x[3, 1] = float('nan')
    x[M - 1, I - 1] = float('nan')
    x = x.requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    gy = torch.randn(M, O, device=dev)
    assert _lib.select_gemm_tile(tile) == 0
    try:
        y = L.linear(lin, x, activation='relu')
        (y * gy).sum().backward()
        gx = x.grad.clone()
        lin.weight.grad = lin.bias.grad = None
        y2 = torch.relu(L.linear(lin, x2))
        (y2 * gy).sum().backward()
    finally:
        _lib.select_gemm_tile(-1)
    nan = torch.isnan(y2)
    assert bool(nan[3].all()) and bool(nan[M - 1].all()) and int(nan.sum()) == 2 * O      # the NaN rows, nothing else
    assert torch.equal(torch.isnan(y), nan)
    assert torch.equal(y[~nan], y2[~nan])
    # rows without a NaN: gradients bit for bit; NaN rows: NaN gradients in both forms (g passes the mask, g W carries the NaN on)
    ok = torch.ones(M, dtype=torch.bool, device=dev)
    ok[3] = ok[M - 1] = False
    assert torch.equal(gx[ok], x2.grad[ok])
    assert torch.equal(torch.isnan(gx), torch.isnan(x2.grad))
    assert not bool(torch.isfinite(y.sum()))          # what the loss, and with it the Trainer's check, gets to see
