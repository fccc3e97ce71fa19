"""Single-kernel parity on the GPU, through the C ABI (vits_op_*), against plain PyTorch fp32 on the CPU.

Tolerances: fp32 MFMA accumulates in a different order than the CPU; with |values| ~ 1 and K <= 8448 products the
difference stays below 2e-4 absolute (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def N():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from vits_amd import _native
    return _native


def dev(t):
    return None if t is None else t.cuda().contiguous()


PREC = {"fp32": 0, "bf16x3": 1, "bf16": 2, "f16x2": 3}


@pytest.fixture(params=["fp32", "bf16x3", "bf16", "f16x2"])
def prec(request):
    return PREC[request.param]


def rq(t, prec):
    """A conv operand as the kernel multiplies it: the plain bf16 mode rounds weights and (after the input prologue)
    activations to nearest-even bf16; products of two bf16 values are exact in fp32, so the CPU reference on rounded
    operands differs from the kernel by accumulation order only and the fp32-grade tolerance applies unchanged."""
    return t.to(torch.bfloat16).to(torch.float32) if prec == PREC["bf16"] else t


def gpu_conv(N, w, bias, x, dilation=1, mode=0, flags=0, lengths=None, cbias=None, in_cbias=None, res=None,
             out_init=None, in_scale=1.0, in_slope=1.0, up=1, bias_rows=None, prec=0):
    """w: logical [rows, cin, taps] (already in kernel row order for transposed convs)."""
    from vits_amd import weights
    rows, cin, taps = w.shape
    B, _, L = x.shape
    cout = rows // 2 if mode == N.MODE_GATE else rows // up
    pk = dev(weights.pack_rows(w, prec))
    b = dev(bias_rows if bias_rows is not None else bias)
    xd, ld, cb, icb, rd = dev(x), dev(lengths), dev(cbias), dev(in_cbias), dev(res)
    out = dev(out_init) if out_init is not None else torch.full((B, cout, L * up), float("nan"), device="cuda")
    N.check(N.lib.vits_op_conv1d(pk.data_ptr(), N.ptr(b), cin, rows, taps, dilation, mode, prec, xd.data_ptr(), B, L,
                                 N.ptr(ld), N.ptr(cb), N.ptr(icb), N.ptr(rd), out.data_ptr(), flags, in_scale,
                                 in_slope, torch.cuda.current_stream().cuda_stream), "op_conv1d")
    torch.cuda.synchronize()
    return out.cpu()


CONV_CASES = [
    # cout, cin, k, dil, B, L        (short launches run on the split-K kernel in bf16x3 mode, long ones on the one-role
    #                                 kernel; the comments name the one-role tile they select)
    (64, 32, 3, 1, 2, 300),       # cin < 64: one-role <2,2,1,4>
    (256, 256, 11, 5, 1, 700),    # widest receptive field (halo 50)
    (32, 32, 7, 3, 2, 1500),      # 32 rows: <1,2,1,4> (fp32 mode: <1,4,1,4>)
    (32, 64, 3, 1, 1, 250),       # <1,2,1,4>
    (96, 192, 1, 1, 2, 70),       # <1,1,1,4>
    (768, 192, 3, 1, 2, 50),      # short sequence
    (192, 768, 3, 1, 3, 37),      # deep reduction, 48 channel chunks
    (576, 192, 1, 1, 2, 258),
    (512, 192, 7, 1, 1, 258),
    (64, 16, 5, 1, 1, 1),         # single time step
    (64, 64, 7, 3, 4, 20000),     # long: one-role <2,2,1,4>
    (128, 128, 3, 1, 4, 9000),    # long: one-role <2,2,2,2>
    (256, 256, 3, 1, 4, 4500),    # long: one-role <2,2,4,1>
    (256, 256, 7, 1, 4, 4500),    # long and deep: split-K with many blocks per CU
]


@pytest.mark.parametrize("cout,cin,k,dil,B,L", CONV_CASES)
def test_conv_plain(N, prec, cout, cin, k, dil, B, L):
    g = torch.Generator().manual_seed(cout * 131 + cin + k + L)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    ref = F.conv1d(rq(x, prec), rq(w, prec), b, padding=(k - 1) // 2 * dil, dilation=dil)
    got = gpu_conv(N, w, b, x, dilation=dil, prec=prec)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("C,k,dil,L", [(64, 7, 3, 12000), (128, 3, 1, 6000), (256, 11, 1, 3000)])
def test_conv_f16x2_small_and_large_grids_agree(N, C, k, dil, L):
    """The f16x2 kernel has two builds of its 64-row-tile configurations: one for grids that fill the chip (three
    blocks per CU) and one for small grids (weight fragments of the next taps kept in flight in registers).  A batch of
    16 goes to the first, its rows one at a time to the second; the bytes must be the same, and right against fp32."""
    p = PREC["f16x2"]
    g = torch.Generator().manual_seed(C + k + L)
    B = 16
    w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    b = torch.randn(C, generator=g)
    x = torch.randn(B, C, L, generator=g)
    whole = gpu_conv(N, w, b, x, dilation=dil, prec=p)
    for i in (0, 7, 15):
        row = gpu_conv(N, w, b, x[i:i + 1], dilation=dil, prec=p)
        assert torch.equal(row[0], whole[i])
    ref = F.conv1d(x[:2], w, b, padding=(k - 1) // 2 * dil, dilation=dil)
    assert (whole[:2] - ref).abs().max() < 2e-5


def test_conv_prologue_epilogue_flags(N, prec):
    g = torch.Generator().manual_seed(7)
    cout, cin, k, B, L = 64, 48, 3, 3, 333
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    lengths = torch.tensor([333, 200, 17], dtype=torch.int32)
    m = (torch.arange(L)[None, :] < lengths[:, None]).float()[:, None, :]
    cbias = torch.randn(B, cout, generator=g)
    icb = torch.randn(B, cin, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    init = torch.randn(B, cout, L, generator=g)
    # input: ((x + in_cbias) * scale) -> lrelu -> mask ; output: relu -> mask -> + res -> + out -> mask
    xin = F.leaky_relu((x + icb[:, :, None]) * 0.5, 0.1) * m
    y = F.conv1d(rq(xin, prec), rq(w, prec), b, padding=1) + cbias[:, :, None]
    ref = ((torch.relu(y) * m) + res + init) * m
    flags = N.F_IN_LRELU | N.F_IN_MASK | N.F_RELU | N.F_MASK_PRE | N.F_RES | N.F_ACCUM | N.F_MASK_POST
    got = gpu_conv(N, w, b, x, flags=flags, lengths=lengths, cbias=cbias, in_cbias=icb, res=res, out_init=init,
                   in_scale=0.5, in_slope=0.1, prec=prec)
    assert (got - ref).abs().max() < 2e-5
    # res - y form used by the coupling layer
    ref2 = (res - (F.conv1d(rq(x, prec), rq(w, prec), b, padding=1) * m)) * m
    got2 = gpu_conv(N, w, b, x, flags=N.F_MASK_PRE | N.F_RES_SUB | N.F_MASK_POST, lengths=lengths, res=res, prec=prec)
    assert (got2 - ref2).abs().max() < 2e-5


@pytest.mark.parametrize("L,B", [(40, 2), (258, 2), (700, 2), (9000, 4)])     # the last one: one-role gate kernel
def test_conv_gate(N, prec, L, B):
    g = torch.Generator().manual_seed(L)
    H, k = 192, 5
    w = torch.randn(2 * H, H, k, generator=g) / math.sqrt(H * k)
    b = torch.randn(2 * H, generator=g)
    x = torch.randn(B, H, L, generator=g)
    cb = torch.randn(B, 2 * H, generator=g)
    a = F.conv1d(rq(x, prec), rq(w, prec), b, padding=2) + cb[:, :, None]
    ref = torch.tanh(a[:, :H]) * torch.sigmoid(a[:, H:])
    got = gpu_conv(N, w, b, x, mode=N.MODE_GATE, cbias=cb, prec=prec)
    assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("u,k,cin,cout,L", [(8, 16, 512, 256, 258), (8, 16, 256, 128, 1500), (4, 4, 128, 64, 700),
                                            (2, 4, 64, 32, 1111), (8, 16, 32, 16, 3)])
def test_conv_transpose(N, prec, u, k, cin, cout, L):
    from vits_amd import weights
    g = torch.Generator().manual_seed(u * 1000 + L)
    wt = torch.randn(cin, cout, k, generator=g) / math.sqrt(cin * k / u)
    b = torch.randn(cout, generator=g)
    x = torch.randn(2, cin, L, generator=g)
    if cout * u % 64:
        pytest.skip("rows must be a multiple of 64")
    # x * float32(1/3) as the kernel's in_scale does (x / 3 differs in the last bit, which can flip a bf16 rounding)
    ref = F.conv_transpose1d(rq(F.leaky_relu(x * (1.0 / 3), 0.1), prec), rq(wt, prec), b, stride=u, padding=(k - u) // 2)
    w_rows, b_rows, mode, taps = weights.transposed_conv_rows(wt, b, u, k, (k - u) // 2)
    got = gpu_conv(N, w_rows, None, x, mode=mode, up=u, bias_rows=b_rows, flags=N.F_IN_LRELU, in_scale=1.0 / 3,
                   in_slope=0.1, prec=prec)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("pname", ["bf16", "f16x2"])
@pytest.mark.parametrize("u,k,cin,cout,L,B", [(8, 16, 512, 256, 258, 2), (8, 16, 256, 128, 1500, 2), (4, 4, 128, 64, 700, 3),
                                              (8, 16, 64, 32, 33, 1), (4, 4, 64, 16, 1, 2)])
def test_conv_transpose_writes_the_staged_copy(N, pname, u, k, cin, cout, L, B):
    """The x8 / x4 transposed convs of the generator also write their result in the staged operand format of the
    ResBlock convs that read it (leaky-relu(0.1), then bf16 or the fp16 h / l split): the fp32 result must be the
    bytes of the launch without the copy, and the copy must decode to exactly the converted fp32 result."""
    from vits_amd import weights
    p = PREC[pname]
    g = torch.Generator().manual_seed(u * 1000 + L + cin)
    wt = torch.randn(cin, cout, k, generator=g) / math.sqrt(cin * k / u)
    b = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, L, generator=g)
    w_rows, b_rows, mode, taps = weights.transposed_conv_rows(wt, b, u, k, (k - u) // 2)
    pk, bd, xd = dev(weights.pack_rows(w_rows, p)), dev(b_rows), dev(x)
    stream = torch.cuda.current_stream().cuda_stream
    Lo = L * u
    plain = torch.full((B, cout, Lo), float("nan"), device="cuda")
    both = torch.full((B, cout, Lo), float("nan"), device="cuda")
    staged = torch.full((B, cout, Lo), float("nan"), device="cuda")        # an fp32-sized buffer is always enough

    def run(out, out2):
        N.check(N.lib.vits_op_conv1d_out2(pk.data_ptr(), bd.data_ptr(), cin, cout * u, taps, 1, mode, p, xd.data_ptr(), B, L,
                                          0, out.data_ptr(), N.ptr(out2), N.F_IN_LRELU, 1.0 / 3, 0.1, 0.1, stream), "op_conv1d_out2")
    run(plain, None)
    run(both, staged)
    torch.cuda.synchronize()
    assert torch.equal(both, plain)
    act = F.leaky_relu(plain, 0.1).cpu()
    flat = staged.reshape(-1).cpu()
    if pname == "bf16":
        got = flat[:B * cout * Lo // 2].contiguous().view(torch.bfloat16).reshape(B, cout // 8, Lo, 8)
        assert torch.equal(got.permute(0, 1, 3, 2).reshape(B, cout, Lo).float(), act.to(torch.bfloat16).float())
    else:
        got = flat.contiguous().view(torch.float16).reshape(B, cout // 8, 2, Lo, 8)
        h = got[:, :, 0].permute(0, 1, 3, 2).reshape(B, cout, Lo).float()
        lo = got[:, :, 1].permute(0, 1, 3, 2).reshape(B, cout, Lo).float()
        assert torch.equal(h, act.to(torch.float16).float())
        assert torch.equal(lo, (act - h).to(torch.float16).float())


def test_conv_out2_is_refused_where_no_kernel_writes_it(N):
    """A second copy is asked of a conv no kernel writes one for (x2 transposed conv, fp32 arithmetic): an error, not a
    silently missing tensor."""
    from vits_amd import weights
    g = torch.Generator().manual_seed(5)
    stream = torch.cuda.current_stream().cuda_stream
    wt = torch.randn(64, 32, 4, generator=g)
    w_rows, b_rows, mode, taps = weights.transposed_conv_rows(wt, torch.zeros(32), 2, 4, 1)
    x = torch.randn(1, 64, 100, device="cuda")
    out, out2 = torch.empty(1, 32, 200, device="cuda"), torch.empty(1, 32, 200, device="cuda")
    for p, w, md, rows, tp in ((PREC["f16x2"], w_rows, mode, 64, taps), (PREC["fp32"], torch.randn(64, 64, 3, generator=g), 0, 64, 3)):
        pk = dev(weights.pack_rows(w, p))
        rc = N.lib.vits_op_conv1d_out2(pk.data_ptr(), 0, 64, rows, tp, 1, md, p, x.data_ptr(), 1, 100, 0, out.data_ptr(),
                                       out2.data_ptr(), 0, 1.0, 1.0, 0.1, stream)
        assert rc != 0


@pytest.mark.parametrize("pname,bound", [("fp32", 2e-6), ("bf16x3", 2e-6), ("f16x2", 2e-6)])
def test_conv_split_schemes_over_a_wide_dynamic_range(N, pname, bound):
    """The operand splits must hold their relative accuracy for small and large values alike (fp16 subnormals carry
    the low term of small operands): inputs spread log-uniformly over 1e-4 .. 1e2, error measured against fp64 and
    normalised by sum |w| |x|, the scale a per-product relative error acts on.  2^-22 per product (f16x2) plus fp32
    accumulation stays below 2e-6 of that scale."""
    g = torch.Generator().manual_seed(123)
    C, k, L, B = 64, 7, 4000, 2
    w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    x = torch.randn(B, C, L, generator=g) * 10.0 ** (torch.rand(B, C, L, generator=g) * 6 - 4)
    b = torch.zeros(C)
    got = gpu_conv(N, w, b, x, prec=PREC[pname]).double()
    ref = F.conv1d(x.double(), w.double(), padding=3)
    scale = F.conv1d(x.double().abs(), w.double().abs(), padding=3)
    assert float(((got - ref).abs() / scale).max()) <= bound


@pytest.mark.parametrize("pname", ["bf16", "f16x2"])
@pytest.mark.parametrize("C,k,dil,L,B", [(64, 7, 3, 3000, 2), (128, 3, 1, 700, 3), (32, 11, 5, 1500, 2), (256, 3, 1, 90, 2)])
def test_conv_staged_pair(N, pname, C, k, dil, L, B):
    """A ResBlock pair with the tensor between the two convs in the second conv's staged operand format: the first
    conv writes leaky-relu(result) converted (bf16: [C/8][L][8]; f16x2: [C/8][h, l][L][8] fp16), the second copies it
    into LDS as is.  The values are those the unstaged pair converts at staging, so the final result is bit-identical;
    the staged bytes are decoded and checked too."""
    from vits_amd import weights
    p = PREC[pname]
    g = torch.Generator().manual_seed(C + k + L)
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    x = torch.randn(B, C, L, generator=g)
    pk1, pk2 = dev(weights.pack_rows(w1, p)), dev(weights.pack_rows(w2, p))
    xd, b1d, b2d = dev(x), dev(b1), dev(b2)
    stream = torch.cuda.current_stream().cuda_stream

    def conv(pk, bias, src, dst, dilation, flags, res=None):
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), bias.data_ptr(), C, C, k, dilation, 0, p, src.data_ptr(), B, L, 0, 0,
                                     0, N.ptr(res), dst.data_ptr(), flags, 1.0, 0.1, stream), "op_conv1d")

    t_plain = torch.empty(B, C, L, device="cuda")
    conv(pk1, b1d, xd, t_plain, dil, N.F_IN_LRELU)
    out_plain = torch.empty(B, C, L, device="cuda")
    conv(pk2, b2d, t_plain, out_plain, 1, N.F_IN_LRELU | N.F_RES, res=xd)
    staged = torch.full((B, C, L), float("nan"), device="cuda")           # an fp32-sized buffer is always enough
    conv(pk1, b1d, xd, staged, dil, N.F_IN_LRELU | N.F_OUT_STAGED)
    out_staged = torch.empty(B, C, L, device="cuda")
    conv(pk2, b2d, staged, out_staged, 1, N.F_IN_STAGED | N.F_RES, res=xd)
    torch.cuda.synchronize()
    assert torch.equal(out_staged, out_plain)
    act = F.leaky_relu(t_plain, 0.1).cpu()
    flat = staged.reshape(-1).cpu()                    # batch rows are packed back to back in the staged element size
    if pname == "bf16":
        got = flat[:B * C * L // 2].contiguous().view(torch.bfloat16).reshape(B, C // 8, L, 8)
        assert torch.equal(got.permute(0, 1, 3, 2).reshape(B, C, L).float(), act.to(torch.bfloat16).float())
    else:
        got = flat.contiguous().view(torch.float16).reshape(B, C // 8, 2, L, 8)
        h = got[:, :, 0].permute(0, 1, 3, 2).reshape(B, C, L).float()
        lo = got[:, :, 1].permute(0, 1, 3, 2).reshape(B, C, L).float()
        assert torch.equal(h, act.to(torch.float16).float())
        assert torch.equal(lo, (act - h).to(torch.float16).float())


@pytest.mark.parametrize("C,k,dil,L,kind", [(128, 11, 5, 16512, "res"), (128, 3, 1, 16400, "staged_out"),
                                            (256, 7, 3, 2064, "res"), (256, 11, 1, 2000, "plain"),
                                            (64, 3, 5, 9000, "res"), (64, 7, 9, 8190, "staged_out"),
                                            (64, 11, 5, 8193, "res_acc")])
def test_conv_lds_dma_matches_register_kernel(N, C, k, dil, L, kind):
    """Staged-input f16x2 convs whose grid fills the chip run on the kernels that bring weights and activations into
    LDS by DMA (128 x 256, 128 x 128 and 64 x 256 block tiles); a single batch row is a small grid and stays with the
    register-staging kernel.  A batch of 16 is compared with its rows one at a time -- at both sequence ends (the DMA's
    out-of-range lanes are the zero padding), across tile borders, for every epilogue the generator uses (residual,
    residual + accumulate, staged output, plain).  3 taps: the 32 x 32 x 16 build, same operands in the same order, the
    same bytes.  7 and 11 taps: the 16 x 16 x 32 build sums the same products 32 at a time instead of 16, so the
    results agree to fp32 summation order (1e-5 absolute on sums of up to 2816 products of magnitude 1..5: a few ulps)
    rather than bit for bit."""
    from vits_amd import weights
    p = PREC["f16x2"]
    g = torch.Generator().manual_seed(C * 31 + k * 7 + L)
    B = 16
    w0 = torch.randn(C, C, 3, generator=g) / math.sqrt(C * 3)
    w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    pk0, pk = dev(weights.pack_rows(w0, p)), dev(weights.pack_rows(w, p))
    b0, b = dev(torch.randn(C, generator=g)), dev(torch.randn(C, generator=g))
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    stream = torch.cuda.current_stream().cuda_stream
    flags = N.F_IN_STAGED | {"res": N.F_RES, "res_acc": N.F_RES | N.F_ACCUM, "staged_out": N.F_OUT_STAGED, "plain": 0}[kind]

    def run(xr, out):
        n = xr.shape[0]
        staged = torch.empty(n, C, L, device="cuda")
        N.check(N.lib.vits_op_conv1d(pk0.data_ptr(), b0.data_ptr(), C, C, 3, 1, 0, p, xr.data_ptr(), n, L, 0, 0, 0, 0,
                                     staged.data_ptr(), N.F_IN_LRELU | N.F_OUT_STAGED, 1.0, 0.1, stream), "first conv")
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), b.data_ptr(), C, C, k, dil, 0, p, staged.data_ptr(), n, L, 0, 0, 0,
                                     xr.data_ptr() if flags & N.F_RES else 0, out.data_ptr(), flags, 1.0, 0.1, stream),
                "second conv")
        return out

    def values(t):
        """fp32 values of one batch row of the result (a staged result is fp16 h / l planes [C/8][2][L][8])."""
        if kind != "staged_out":
            return t
        hl = t.contiguous().view(torch.float16).reshape(C // 8, 2, L, 8).float()
        return (hl[:, 0] + hl[:, 1]).permute(0, 2, 1).reshape(C, L)

    whole = run(x, init.clone())
    for i in (0, 9, 15):
        row = run(x[i:i + 1].contiguous(), init[i:i + 1].clone())
        torch.cuda.synchronize()
        if k == 3:
            assert torch.equal(row[0].view(torch.int32), whole[i].view(torch.int32))  # bytes (staged output is fp16 pairs)
        else:
            assert (values(row[0]) - values(whole[i])).abs().max() < 1e-5
    if kind != "staged_out":
        xc = x[:2].cpu()
        t = F.leaky_relu(F.conv1d(F.leaky_relu(xc, 0.1), w0, b0.cpu(), padding=1), 0.1)
        ref = F.conv1d(t, w, b.cpu(), padding=(k - 1) // 2 * dil, dilation=dil)
        if flags & N.F_RES:
            ref = ref + xc
        if flags & N.F_ACCUM:
            ref = ref + init[:2].cpu()
        assert (whole[:2].cpu() - ref).abs().max() < 5e-5


@pytest.mark.parametrize("C,k,dil,L,kind", [(128, 11, 5, 16512, "res"), (128, 3, 1, 16400, "res_acc"),
                                            (256, 7, 3, 2064, "res"), (256, 11, 1, 2000, "plain"),
                                            (128, 7, 9, 4100, "staged_out"), (256, 3, 5, 8193, "res")])
def test_conv_bf16_lds_dma_matches_register_kernel(N, C, k, dil, L, kind):
    """Staged-input plain-bf16 convs with rows in 128s whose grid fills the chip run on the one-plane LDS-DMA kernel
    (128 x 256 and 128 x 128 block tiles); a single batch row is a small grid and stays with the register-staging
    kernel.  A batch of 16 against its rows one at a time, at both sequence ends and across tile borders, for the
    epilogues the generator uses.  The same bf16 products, summed tap by tap over a pair of 16-channel chunks instead
    of chunk by chunk: equal to fp32 summation order (1e-5 absolute on sums of magnitude 1..5; a staged bf16 result may
    differ by one rounding step where a sum sits on a rounding boundary)."""
    from vits_amd import weights
    p = PREC["bf16"]
    g = torch.Generator().manual_seed(C * 31 + k * 7 + L)
    B = 16
    w0 = torch.randn(C, C, 3, generator=g) / math.sqrt(C * 3)
    w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    pk0, pk = dev(weights.pack_rows(w0, p)), dev(weights.pack_rows(w, p))
    b0, b = dev(torch.randn(C, generator=g)), dev(torch.randn(C, generator=g))
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    stream = torch.cuda.current_stream().cuda_stream
    flags = N.F_IN_STAGED | {"res": N.F_RES, "res_acc": N.F_RES | N.F_ACCUM, "staged_out": N.F_OUT_STAGED, "plain": 0}[kind]

    def run(xr, out):
        n = xr.shape[0]
        staged = torch.empty(n, C, L, device="cuda")
        N.check(N.lib.vits_op_conv1d(pk0.data_ptr(), b0.data_ptr(), C, C, 3, 1, 0, p, xr.data_ptr(), n, L, 0, 0, 0, 0,
                                     staged.data_ptr(), N.F_IN_LRELU | N.F_OUT_STAGED, 1.0, 0.1, stream), "first conv")
        N.check(N.lib.vits_op_conv1d(pk.data_ptr(), b.data_ptr(), C, C, k, dil, 0, p, staged.data_ptr(), n, L, 0, 0, 0,
                                     xr.data_ptr() if flags & N.F_RES else 0, out.data_ptr(), flags, 1.0, 0.1, stream),
                "second conv")
        return out

    def values(t):
        """fp32 values of one batch row of the result (a staged result is bf16 [C/8][L][8] in the row's first half)."""
        if kind != "staged_out":
            return t
        v = t.contiguous().reshape(-1)[:C * L // 2].contiguous().view(torch.bfloat16).reshape(C // 8, L, 8).float()
        return v.permute(0, 2, 1).reshape(C, L)

    whole = run(x, init.clone())
    if kind == "staged_out":                            # rows of a staged batch are packed back to back, 2 bytes a value
        whole_rows = whole.reshape(-1).view(torch.bfloat16)[:B * C * L].reshape(B, C // 8, L, 8).float()
    for i in (0, 9, 15):
        row = run(x[i:i + 1].contiguous(), init[i:i + 1].clone())
        torch.cuda.synchronize()
        if kind == "staged_out":
            d = (values(row[0]) - whole_rows[i].permute(0, 2, 1).reshape(C, L)).abs()
            assert d.max() <= 0.04 and (d > 0).float().mean() < 1e-3      # isolated one-step bf16 differences only
        else:
            assert (row[0] - whole[i]).abs().max() < 1e-5
    if kind != "staged_out":
        xc = x[:2].cpu()
        t = rq(F.leaky_relu(F.conv1d(rq(F.leaky_relu(xc, 0.1), p), rq(w0, p), b0.cpu(), padding=1), 0.1), p)
        ref = F.conv1d(t, rq(w, p), b.cpu(), padding=(k - 1) // 2 * dil, dilation=dil)
        if flags & N.F_RES:
            ref = ref + xc
        if flags & N.F_ACCUM:
            ref = ref + init[:2].cpu()
        assert (whole[:2].cpu() - ref).abs().max() < 2e-2       # a bf16 rounding of the intermediate can flip on its last bit


@pytest.mark.parametrize("C,k,dil,L,B,acc", [(32, 3, 1, 1000, 2, False), (32, 7, 3, 3000, 2, True),
                                             (32, 11, 5, 2500, 3, False), (32, 11, 1, 224, 1, True),
                                             (32, 3, 5, 225, 2, False), (32, 7, 5, 50, 2, False),
                                             (64, 3, 1, 1000, 2, True), (64, 7, 5, 97, 3, False), (64, 11, 3, 96, 1, False),
                                             (32, 5, 3, 700, 2, True), (32, 9, 1, 449, 2, False),
                                             (64, 5, 5, 193, 2, False), (64, 9, 3, 300, 1, True)])
def test_resblock_pair_fused(N, C, k, dil, L, B, acc):
    """The fused ResBlock pair (32 or 64 channels; tensor between the convs kept in LDS) against
    the two separate f16x2 convs: same operands in the same order, so the same bytes -- across tile borders (224- and
    96-step tiles), at both sequence ends
    (zero padding of the second conv's input) and with the accumulate form used for the sum of the three chains."""
    from vits_amd import weights
    p = PREC["f16x2"]
    g = torch.Generator().manual_seed(k * 100 + dil + L)
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    b1, b2 = dev(torch.randn(C, generator=g)), dev(torch.randn(C, generator=g))
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    pk1, pk2 = dev(weights.pack_rows(w1, p)), dev(weights.pack_rows(w2, p))
    stream = torch.cuda.current_stream().cuda_stream
    t = torch.empty(B, C, L, device="cuda")
    N.check(N.lib.vits_op_conv1d(pk1.data_ptr(), b1.data_ptr(), C, C, k, dil, 0, p, x.data_ptr(), B, L, 0, 0, 0, 0,
                                 t.data_ptr(), N.F_IN_LRELU, 1.0, 0.1, stream), "conv1")
    ref = init.clone()
    N.check(N.lib.vits_op_conv1d(pk2.data_ptr(), b2.data_ptr(), C, C, k, 1, 0, p, t.data_ptr(), B, L, 0, 0, 0,
                                 x.data_ptr(), ref.data_ptr(), N.F_IN_LRELU | N.F_RES | (N.F_ACCUM if acc else 0), 1.0,
                                 0.1, stream), "conv2")
    out = init.clone()
    N.check(N.lib.vits_op_resblock_pair(pk1.data_ptr(), b1.data_ptr(), pk2.data_ptr(), b2.data_ptr(), C, k, dil,
                                        x.data_ptr(), out.data_ptr(), B, L, 1 if acc else 0, 0.1, stream), "pair")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    # and against plain fp32 math
    xc = x.cpu()
    y = F.conv1d(F.leaky_relu(xc, 0.1), w1, b1.cpu(), padding=(k - 1) // 2 * dil, dilation=dil)
    y = F.conv1d(F.leaky_relu(y, 0.1), w2, b2.cpu(), padding=(k - 1) // 2) + xc + (init.cpu() if acc else 0)
    assert (out.cpu() - y).abs().max() < 5e-5


@pytest.mark.parametrize("C,k,dil,L,B,acc", [(32, 3, 1, 1000, 2, False), (32, 7, 3, 3000, 2, True),
                                             (32, 11, 5, 2500, 3, False), (32, 11, 1, 224, 1, True),
                                             (64, 3, 1, 1000, 2, True), (64, 7, 5, 97, 3, False), (64, 11, 3, 500, 2, True),
                                             # 64 channels: 224-step tiles, lengths on and around a tile boundary
                                             (64, 3, 5, 224, 1, False), (64, 11, 1, 225, 2, False),
                                             (64, 5, 3, 1345, 2, True), (64, 9, 1, 449, 1, True)])
def test_resblock_pair_fused_bf16(N, C, k, dil, L, B, acc):
    """The fused pair in the plain-bf16 arithmetic (what precision="bf16" launches for the 32- and 64-channel stages)
    against the two separate bf16 convs, with and without the staged tensor between them: the same roundings, the same
    order of additions, so the same bytes; and against fp32 math on bf16-rounded operands."""
    from vits_amd import weights
    p = PREC["bf16"]
    g = torch.Generator().manual_seed(k * 100 + dil + L + 7)
    w1 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    w2 = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    b1, b2 = dev(torch.randn(C, generator=g)), dev(torch.randn(C, generator=g))
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    pk1, pk2 = dev(weights.pack_rows(w1, p)), dev(weights.pack_rows(w2, p))
    stream = torch.cuda.current_stream().cuda_stream
    t = torch.empty(B, C, L, device="cuda")
    N.check(N.lib.vits_op_conv1d(pk1.data_ptr(), b1.data_ptr(), C, C, k, dil, 0, p, x.data_ptr(), B, L, 0, 0, 0, 0,
                                 t.data_ptr(), N.F_IN_LRELU, 1.0, 0.1, stream), "conv1")
    ref = init.clone()
    N.check(N.lib.vits_op_conv1d(pk2.data_ptr(), b2.data_ptr(), C, C, k, 1, 0, p, t.data_ptr(), B, L, 0, 0, 0,
                                 x.data_ptr(), ref.data_ptr(), N.F_IN_LRELU | N.F_RES | (N.F_ACCUM if acc else 0), 1.0,
                                 0.1, stream), "conv2")
    out = init.clone()
    N.check(N.lib.vits_op_resblock_pair_prec(pk1.data_ptr(), b1.data_ptr(), pk2.data_ptr(), b2.data_ptr(), C, k, dil, p,
                                             x.data_ptr(), out.data_ptr(), B, L, 1 if acc else 0, 0.1, stream), "pair")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    xc = x.cpu()
    y = F.conv1d(rq(F.leaky_relu(xc, 0.1), p), rq(w1, p), b1.cpu(), padding=(k - 1) // 2 * dil, dilation=dil)
    y = F.conv1d(rq(F.leaky_relu(y, 0.1), p), rq(w2, p), b2.cpu(), padding=(k - 1) // 2) + xc + (init.cpu() if acc else 0)
    assert (out.cpu() - y).abs().max() < 2e-2          # a bf16 rounding of the intermediate can flip on its last bit


def _ptr_array(tensors):
    import ctypes as C
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


@pytest.mark.parametrize("prec,C,k,dils,L,B,acc", [
    ("f16x2", 32, 3, (1, 3, 5), 1000, 2, False), ("f16x2", 32, 3, (1, 3, 5), 232, 1, True),
    ("f16x2", 32, 3, (1, 3, 5), 233, 2, True), ("f16x2", 64, 3, (1, 3, 5), 700, 2, True),
    ("f16x2", 64, 3, (1, 3, 5), 104, 1, False), ("f16x2", 32, 7, (1, 3, 5), 1500, 2, True),
    ("f16x2", 32, 11, (1, 3, 5), 700, 1, False), ("f16x2", 64, 7, (1, 3, 5), 333, 1, False),
    ("f16x2", 32, 3, (1, 2), 500, 2, False), ("f16x2", 32, 5, (1, 3, 5, 2), 40, 3, True),
    ("f16x2", 64, 11, (3,), 300, 2, True), ("bf16", 64, 11, (1, 3), 700, 1, False),
    ("bf16", 32, 3, (1, 3, 5), 1000, 2, True), ("bf16", 32, 7, (1, 3, 5), 1761, 2, False),
    ("bf16", 32, 11, (1, 3, 5), 1200, 1, True), ("bf16", 64, 3, (1, 3, 5), 465, 2, True),
    ("bf16", 64, 7, (1, 3, 5), 900, 2, False), ("bf16", 64, 11, (1, 3, 5), 300, 1, True),
    ("bf16", 32, 7, (1, 3, 5), 440, 1, True), ("bf16", 64, 3, (5, 1, 3), 17, 2, False)])
def test_resblock_chain_fused(N, prec, C, k, dils, L, B, acc):
    """A whole ResBlock in one launch (the carrier of a time tile stays in registers across its pairs) against the
    same pairs launched one after the other: the same products in the same order, so the same bytes -- across tile
    borders (lengths on and around a tile's 232 / 104 / 184 / 440 owned steps), at both sequence ends, shorter than one
    tile, with two, three and four pairs, and with the accumulate form used for the sum of a stage's ResBlocks; and
    against plain fp32 math (reference modules.py:206-219)."""
    from vits_amd import weights
    p = PREC[prec]
    g = torch.Generator().manual_seed(k * 100 + L + C)
    n = len(dils)
    ws = [(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), torch.randn(C, C, k, generator=g) / math.sqrt(C * k))
          for _ in range(n)]
    bs = [(dev(torch.randn(C, generator=g) * 0.3), dev(torch.randn(C, generator=g) * 0.3)) for _ in range(n)]
    pk = [(dev(weights.pack_rows(w1, p)), dev(weights.pack_rows(w2, p))) for w1, w2 in ws]
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    stream = torch.cuda.current_stream().cuda_stream
    cur = x
    for q in range(n):                                     # the pair-by-pair path: ping-pong carriers, the last one joins the sum
        last = q + 1 == n
        dst = init.clone() if last else torch.empty_like(x)
        N.check(N.lib.vits_op_resblock_pair_prec(pk[q][0].data_ptr(), bs[q][0].data_ptr(), pk[q][1].data_ptr(),
                                                 bs[q][1].data_ptr(), C, k, dils[q], p, cur.data_ptr(), dst.data_ptr(), B,
                                                 L, 1 if (acc and last) else 0, 0.1, stream), "pair")
        cur = dst
    ref = cur
    out = init.clone()
    import ctypes as Ct
    N.check(N.lib.vits_op_resblock_chain(_ptr_array([a for a, _ in pk]), _ptr_array([a for a, _ in bs]),
                                         _ptr_array([b for _, b in pk]), _ptr_array([b for _, b in bs]), C, k,
                                         (Ct.c_int * n)(*dils), n, p, x.data_ptr(), out.data_ptr(), B, L, 1 if acc else 0,
                                         0.1, stream), "chain")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    y = x.cpu()
    for q in range(n):
        (w1, w2), (b1, b2) = ws[q], bs[q]
        t = F.conv1d(rq(F.leaky_relu(y, 0.1), p), rq(w1, p), b1.cpu(), padding=(k - 1) // 2 * dils[q], dilation=dils[q])
        y = F.conv1d(rq(F.leaky_relu(t, 0.1), p), rq(w2, p), b2.cpu(), padding=(k - 1) // 2) + y
    if acc:
        y = y + init.cpu()
    assert (out.cpu() - y).abs().max() < (1e-4 if prec == "f16x2" else 5e-2)


@pytest.mark.parametrize("prec", ["bf16", "f16x2"])
@pytest.mark.parametrize("dils,L,B,acc", [((1, 3, 5), 1000, 2, True), ((1, 3, 5), 104, 1, False), ((1, 3, 5), 105, 2, True),
                                           ((1,), 333, 1, False), ((1, 3), 37, 2, True)])
def test_resblock_chain_128_channels(N, prec, dils, L, B, acc):
    """The whole-ResBlock kernel's 128-channel builds (three taps: the generator's second stage) against the same convs
    launched one by one (the same operands, fp32 accumulation chunk by chunk and tap by tap: the same bytes), across
    tile borders (104 owned steps a tile with dilations 1, 3, 5) and with the accumulate form."""
    from vits_amd import weights
    import ctypes as Ct
    p, C, k = PREC[prec], 128, 3
    g = torch.Generator().manual_seed(L + len(dils))
    n = len(dils)
    ws = [(torch.randn(C, C, k, generator=g) / math.sqrt(C * k), torch.randn(C, C, k, generator=g) / math.sqrt(C * k))
          for _ in range(n)]
    bs = [(dev(torch.randn(C, generator=g) * 0.3), dev(torch.randn(C, generator=g) * 0.3)) for _ in range(n)]
    pk = [(dev(weights.pack_rows(w1, p)), dev(weights.pack_rows(w2, p))) for w1, w2 in ws]
    x = dev(torch.randn(B, C, L, generator=g))
    init = dev(torch.randn(B, C, L, generator=g))
    stream = torch.cuda.current_stream().cuda_stream

    def conv(pkw, bias, src, dst, dilation, flags, res=None):
        N.check(N.lib.vits_op_conv1d(pkw.data_ptr(), bias.data_ptr(), C, C, k, dilation, 0, p, src.data_ptr(), B, L, 0, 0, 0,
                                     N.ptr(res), dst.data_ptr(), flags, 1.0, 0.1, stream), "op_conv1d")
    cur = x
    for q in range(n):
        last = q + 1 == n
        t = torch.empty_like(x)
        conv(pk[q][0], bs[q][0], cur, t, dils[q], N.F_IN_LRELU)
        dst = init.clone() if last else torch.empty_like(x)
        conv(pk[q][1], bs[q][1], t, dst, 1, N.F_IN_LRELU | N.F_RES | (N.F_ACCUM if (acc and last) else 0), res=cur)
        cur = dst
    ref = cur
    out = init.clone()
    N.check(N.lib.vits_op_resblock_chain(_ptr_array([a for a, _ in pk]), _ptr_array([a for a, _ in bs]),
                                         _ptr_array([b for _, b in pk]), _ptr_array([b for _, b in bs]), C, k,
                                         (Ct.c_int * n)(*dils), n, p, x.data_ptr(), out.data_ptr(), B, L, 1 if acc else 0,
                                         0.1, stream), "chain")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    y = x.cpu()
    for q in range(n):
        (w1, w2), (b1, b2) = ws[q], bs[q]
        t = F.conv1d(rq(F.leaky_relu(y, 0.1), p), rq(w1, p), b1.cpu(), padding=(k - 1) // 2 * dils[q], dilation=dils[q])
        y = F.conv1d(rq(F.leaky_relu(t, 0.1), p), rq(w2, p), b2.cpu(), padding=(k - 1) // 2) + y
    if acc:
        y = y + init.cpu()
    assert (out.cpu() - y).abs().max() < (1e-4 if prec == "f16x2" else 5e-2)


def test_resblock_chain_refuses_unsupported_shapes(N):
    """Five pairs, halos that add up to more than 64 steps a side, in-place use and other channel or tap counts are
    errors, never launches."""
    import ctypes as Ct
    from vits_amd import weights
    p = PREC["f16x2"]
    stream = torch.cuda.current_stream().cuda_stream

    def call(C, k, dils, in_place=False):
        w = torch.zeros(C, C, k)
        pk, b = dev(weights.pack_rows(w, p)), dev(torch.zeros(C))
        x, out = dev(torch.zeros(1, C, 64)), dev(torch.zeros(1, C, 64))
        n = len(dils)
        return N.lib.vits_op_resblock_chain(_ptr_array([pk] * n), _ptr_array([b] * n), _ptr_array([pk] * n),
                                            _ptr_array([b] * n), C, k, (Ct.c_int * n)(*dils), n, p, x.data_ptr(),
                                            (x if in_place else out).data_ptr(), 1, 64, 0, 0.1, stream)
    assert call(32, 3, (1, 3, 5)) == 0
    assert call(32, 3, (1,)) == 0 and call(32, 3, (1, 1, 1, 1, 1)) != 0
    assert call(32, 11, (5, 5, 5)) != 0 and call(96, 3, (1, 3, 5)) != 0 and call(32, 4, (1, 3, 5)) != 0
    assert call(32, 3, (1, 3, 5), in_place=True) != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("C,k,dil", [(32, 13, 1), (32, 4, 1), (32, 11, 6), (96, 3, 1), (32, 1, 1)])
def test_resblock_pair_refuses_unsupported_shapes(N, C, k, dil):
    """Tap counts without an instantiation (odd, 3..11), a first-conv halo over 25 steps and other channel counts
    are an error, never a launch."""
    from vits_amd import weights
    p = PREC["f16x2"]
    w = torch.zeros(C, C, k)
    pk, b = dev(weights.pack_rows(w, p)), dev(torch.zeros(C))
    x, out = dev(torch.zeros(1, C, 64)), dev(torch.zeros(1, C, 64))
    rc = N.lib.vits_op_resblock_pair(pk.data_ptr(), b.data_ptr(), pk.data_ptr(), b.data_ptr(), C, k, dil,
                                     x.data_ptr(), out.data_ptr(), 1, 64, 0, 0.1, torch.cuda.current_stream().cuda_stream)
    assert rc != 0


@pytest.mark.parametrize("C,L,mask", [(192, 50, False), (256, 333, True), (768, 70, False), (192, 1, True)])
def test_layer_norm(N, C, L, mask):
    g = torch.Generator().manual_seed(C + L)
    B = 3
    x = torch.randn(B, C, L, generator=g) * 3 + 1
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    lengths = torch.tensor([L, max(L // 2, 1), 1], dtype=torch.int32)
    ref = F.layer_norm(x.transpose(1, 2), (C,), gamma, beta, 1e-5).transpose(1, 2)
    if mask:
        ref = ref * (torch.arange(L)[None, :] < lengths[:, None]).float()[:, None, :]
    xd, y = dev(x), torch.empty(B, C, L, device="cuda")
    gd, bd, ld = dev(gamma), dev(beta), dev(lengths)
    N.check(N.lib.vits_op_layer_norm(xd.data_ptr(), y.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                     ld.data_ptr(), B, C, L, int(mask), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max() < 1e-5
    # in place
    N.check(N.lib.vits_op_layer_norm(xd.data_ptr(), xd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                     ld.data_ptr(), B, C, L, int(mask), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (xd.cpu() - ref).abs().max() < 1e-5


def reference_attention(q, k, v, ek, ev, lengths, H, W=4):
    """The reference's formulation with padded relative embeddings and skew reshapes (attentions.py:226-321),
    restated independently of the oracle's banded version."""
    B, C, L = q.shape
    dk = C // H
    q = q.view(B, H, dk, L).transpose(2, 3)
    k = k.view(B, H, dk, L).transpose(2, 3)
    v = v.view(B, H, dk, L).transpose(2, 3)

    def rel_emb(e):
        pad = max(L - (W + 1), 0)
        start = max((W + 1) - L, 0)
        e = F.pad(e, (0, 0, pad, pad))
        return e[start:start + 2 * L - 1]

    def rel_to_abs(x):
        x = F.pad(x, (0, 1)).reshape(B, H, L * 2 * L)
        x = F.pad(x, (0, L - 1))
        return x.view(B, H, L + 1, 2 * L - 1)[:, :, :L, L - 1:]

    def abs_to_rel(x):
        x = F.pad(x, (0, L - 1)).reshape(B, H, L * L + L * (L - 1))
        x = F.pad(x, (L, 0))
        return x.view(B, H, L, 2 * L)[:, :, :, 1:]
    qs = q / math.sqrt(dk)
    scores = qs @ k.transpose(-2, -1) + rel_to_abs(qs @ rel_emb(ek).t())
    m = torch.arange(L)[None, :] < lengths[:, None]
    mask = (m[:, None, :] & m[:, :, None])[:, None]
    scores = scores.masked_fill(~mask, -1e4)
    p = F.softmax(scores, dim=-1)
    out = p @ v + abs_to_rel(p) @ rel_emb(ev)
    return out.transpose(2, 3).reshape(B, C, L)


@pytest.mark.parametrize("L,B", [(3, 2), (12, 2), (50, 2), (130, 2), (255, 2), (256, 2), (258, 2), (600, 2), (861, 2),
                                 (1400, 2), (256, 70), (258, 64), (300, 33), (511, 17)])
def test_attention(N, L, B):
    """Short rows and small grids: the lanes-on-keys kernel; L >= 512, and L >= 256 when the batch fills the chip with
    64-query blocks: the matrix-core kernel (bf16x3 products, one pass over the keys with a running maximum; 64- and
    128-query blocks).  Both against the reference's own formulation (padded relative embeddings, skew reshapes)."""
    g = torch.Generator().manual_seed(L)
    H, dk = 2, 96
    C = H * dk
    qkv = torch.randn(B, 3 * C, L, generator=g)
    ek, ev = torch.randn(9, dk, generator=g) * 0.1, torch.randn(9, dk, generator=g) * 0.1
    lengths = torch.tensor([L, max(1, (2 * L) // 3)] + [max(1, L - 7 * i) for i in range(B - 2)], dtype=torch.int32)
    ref = reference_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ek, ev, lengths, H)
    out = torch.empty(B, C, L, device="cuda")
    qd, kd, vd, ld = dev(qkv), dev(ek), dev(ev), dev(lengths)
    N.check(N.lib.vits_op_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ld.data_ptr(),
                                    out.data_ptr(), B, H, dk, L, 4, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-5


@pytest.mark.parametrize("L,B", [(600, 2), (861, 3), (1400, 2), (256, 70), (258, 64), (511, 17), (2500, 1)])
def test_attention_matrix_core_kernel_in_f16x2(N, L, B):
    """The matrix-core attention with its operands split into fp16 h + l (what the frame-rate encoder runs beside its
    f16x2 convs): the same reference, the same fp32-grade tolerance; and a whole row of masked queries (uniform
    attention, reference attentions.py:240-243 with masked_fill(-1e4)) is reproduced too."""
    g = torch.Generator().manual_seed(L + 1)
    H, dk = 2, 96
    C = H * dk
    qkv = torch.randn(B, 3 * C, L, generator=g) * 1.5
    ek, ev = torch.randn(9, dk, generator=g) * 0.1, torch.randn(9, dk, generator=g) * 0.1
    lengths = torch.tensor([L, max(1, (2 * L) // 3)] + [max(1, L - 7 * i) for i in range(B - 2)], dtype=torch.int32)[:B]
    ref = reference_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ek, ev, lengths, H)
    out = torch.empty(B, C, L, device="cuda")
    qd, kd, vd, ld = dev(qkv), dev(ek), dev(ev), dev(lengths)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    N.lib.vits_op_set_range_flag(flag.data_ptr())
    try:
        N.check(N.lib.vits_op_attention_prec(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ld.data_ptr(), out.data_ptr(), B, H,
                                             dk, L, 4, PREC["f16x2"], torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        N.lib.vits_op_set_range_flag(0)
    assert int(flag.cpu()) == 0
    assert (out.cpu() - ref).abs().max() < 2e-5


def test_attention_f16x2_reports_operands_beyond_the_fp16_range(N):
    g = torch.Generator().manual_seed(3)
    H, dk, L, B = 2, 96, 600, 1
    C = H * dk
    qkv = torch.randn(B, 3 * C, L, generator=g)
    qkv[0, 2 * C + 5, 17] = 1e5                       # one value of v beyond 65504
    ek, ev = torch.zeros(9, dk), torch.zeros(9, dk)
    out = torch.empty(B, C, L, device="cuda")
    qd, kd, vd = dev(qkv), dev(ek), dev(ev)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    N.lib.vits_op_set_range_flag(flag.data_ptr())
    try:
        N.check(N.lib.vits_op_attention_prec(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), 0, out.data_ptr(), B, H, dk, L, 4,
                                             PREC["f16x2"], torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        N.lib.vits_op_set_range_flag(0)
    assert int(flag.cpu()) == 1
