"""GPU parity of the conv-stack primitives against a plain PyTorch fp32 CPU reference of the
same op (floating point: tolerance 1e-4 relative to the tensor's scale, as the north star
states; fp32 MFMA is an exact fp32 FMA chain, so observed errors are ~1e-6)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import fdet_amd
    from fdet_amd import hotpath
    return hotpath


def close(a, b, tol=1e-4):
    a = a.cpu().double(); b = b.cpu().double()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max err {err} vs scale {scale}"


SHAPES = [(3, 64, 60, 60), (5, 64, 15, 15), (4, 64, 30, 30), (2, 32, 30, 30), (2, 8, 7, 9), (1, 16, 20, 20),
          (2, 128, 15, 15), (33, 64, 15, 15),
          # pipelined weight gradient edge cases: one image (odd band count), idle column lanes (W=56), short odd rows
          (1, 64, 60, 56), (2, 64, 9, 13), (7, 64, 6, 60),
          # wide rows (Resnet at 480^2 / 640^2): column-segmented tiles in the bf16x3 kernels
          (1, 64, 12, 240), (1, 32, 9, 320), (2, 64, 20, 100), (1, 16, 6, 164), (2, 64, 17, 80)]


@pytest.mark.parametrize("x3", [False, True], ids=["f32", "bf16x3"])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_fwd_dgrad_wgrad(hp, shape, x3):
    """x3=False: exact fp32 MFMA chain.  x3=True: bf16 hi/lo split on the bf16 matrix cores with
    fp32 accumulation; same 1e-4 tolerance (observed ~1e-5)."""
    N, C, H, W = shape
    if x3 and not hp.x3_supported(C, C):
        pytest.skip("bf16x3 needs channel counts that are multiples of 16")
    if not x3 and W > 250:
        pytest.skip("wide rows are built for the bf16x3 kernels only (the exact-fp32 path stops at ~250 columns)")
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    skip = torch.randn(N, C, H, W, generator=g)
    scale = (torch.rand(N, C, generator=g) > 0.25).float() / 0.75
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w.cuda(), wf, wb, x3=x3)
    xd = x.cuda()
    # forward, conv1 flavour and conv2 flavour
    y_full = torch.full((N, C, H, W), float("nan"), device="cuda")
    y_out = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_fwd(xd, wf, b.cuda(), C, y_full=y_full, skip=skip.cuda(), drop_scale=scale.cuda(), y_out=y_out, x3=x3)
    y1 = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_fwd(xd, wf, b.cuda(), C, y_full=y1, x3=x3)                       # conv1 flavour (own epilogue mode)
    y2 = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_fwd(xd, wf, b.cuda(), C, skip=skip.cuda(), y_out=y2, x3=x3)      # eval block tail
    z = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    close(y_full, z)
    close(y_out, z * scale[:, :, None, None] + skip)
    close(y1, z)
    close(y2, z + skip)
    # data gradient with fused lrelu' and add
    dz = torch.randn(N, C, H, W, generator=g)
    act = torch.randn(N, C, H, W, generator=g)
    add = torch.randn(N, C, H, W, generator=g)
    dx = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_dgrad(dz.cuda(), wb, C, dx, act=act.cuda(), add=add.cuda(), x3=x3)
    ref = F.conv_transpose2d(dz, w, padding=1) * torch.where(act > 0, 1.0, 0.2) + add
    close(dx, ref)
    dxa = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_dgrad(dz.cuda(), wb, C, dxa, act=act.cuda(), x3=x3)
    close(dxa, F.conv_transpose2d(dz, w, padding=1) * torch.where(act > 0, 1.0, 0.2))
    hp.conv3x3_dgrad(dz.cuda(), wb, C, dxa, add=add.cuda(), x3=x3)
    close(dxa, F.conv_transpose2d(dz, w, padding=1) + add)
    # weight gradient
    ws = torch.empty(hp.conv3x3_wgrad_ws_bytes(N, C, C, H, W) // 4, device="cuda")
    dW = torch.full((C, C, 3, 3), float("nan"), device="cuda"); db = torch.full((C,), float("nan"), device="cuda")
    hp.conv3x3_wgrad(xd, dz.cuda(), dW, db, ws, x3=x3 and hp.wgrad_x3_supported(N, C, C, H, W))
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    F.conv2d(xr, wr, br, padding=1).backward(dz)
    close(dW, wr.grad)
    close(db, br.grad)


@pytest.mark.parametrize("cfg", [(3, 4, 64, 15, 15), (2, 2, 64, 60, 60), (16, 2, 64, 15, 15), (2, 3, 64, 30, 30), (2, 2, 32, 20, 20)])
def test_conv3x3_wgrad_batched(hp, cfg):
    """L same-shape layers in ONE launch (the bench path: 2 layers at 60x60, 4 at 30x30, 16 at 15x15) against
    autograd per layer; 1e-4 of the tensor's scale as everywhere."""
    L, N, C, H, W = cfg
    g = torch.Generator().manual_seed(L * 100 + H)
    xs = [torch.randn(N, C, H, W, generator=g) for _ in range(L)]
    dzs = [torch.randn(N, C, H, W, generator=g) for _ in range(L)]
    dWs = [torch.full((C, C, 3, 3), float("nan"), device="cuda") for _ in range(L)]
    dbs = [torch.full((C,), float("nan"), device="cuda") for _ in range(L)]
    ws = torch.empty(hp.conv3x3_wgrad_batched_ws_bytes(L, N, C, C, H, W) // 4, device="cuda")
    hp.conv3x3_wgrad_batched([x.cuda() for x in xs], [d.cuda() for d in dzs], dWs, dbs, ws)
    for l in range(L):
        wr = torch.zeros(C, C, 3, 3, requires_grad=True); br = torch.zeros(C, requires_grad=True)
        F.conv2d(xs[l], wr, br, padding=1).backward(dzs[l])
        close(dWs[l], wr.grad)
        close(dbs[l], br.grad)


@pytest.mark.parametrize("pool", [1, 2])
def test_block_tail(hp, pool):
    g = torch.Generator().manual_seed(pool)
    N, C, H, W = 3, 16, 12, 20
    c = torch.randn(N, C, H, W, generator=g)
    x = torch.randn(N, C, H, W, generator=g)
    x[0, 0, :2, :2] = 1.0; c[0, 0, :2, :2] = 0.0           # a tie inside one window: first max wins
    scale = (torch.rand(N, C, generator=g) > 0.25).float() / 0.75
    out = torch.empty(N, C, H // pool, W // pool, device="cuda")
    hp.block_tail_fwd(c.cuda(), x.cuda(), scale.cuda(), out, pool)
    cr = c.clone().requires_grad_(True); xr = x.clone().requires_grad_(True)
    e = F.leaky_relu(cr, 1.0) * scale[:, :, None, None] + xr
    ref = F.max_pool2d(e, 2) if pool == 2 else e
    assert torch.equal(out.cpu(), ref.detach())
    dout = torch.randn(ref.shape, generator=g)
    # reference for dz2 = d/dz2 where c = lrelu(z2): chain through lrelu'(c)
    z2 = torch.where(c > 0, c, c / 0.2).requires_grad_(True)
    e2 = F.leaky_relu(z2, 0.2) * scale[:, :, None, None] + xr
    r2 = F.max_pool2d(e2, 2) if pool == 2 else e2
    gz, gx = torch.autograd.grad(r2, [z2, xr], dout)
    dz2 = torch.empty(N, C, H, W, device="cuda"); de = torch.empty(N, C, H, W, device="cuda")
    hp.block_tail_bwd(dout.cuda(), c.cuda(), x.cuda(), scale.cuda(), dz2, de if pool == 2 else None, pool)
    assert torch.allclose(dz2.cpu(), gz, rtol=1e-6, atol=1e-6)
    if pool == 2:
        assert torch.equal(de.cpu(), gx)


@pytest.mark.parametrize("cfg", [(2, 64, 480, 10, 8, 2), (3, 8, 480, 10, 8, 2), (2, 32, 240, 3, 2, 1),
                                 (1, 64, 640, 3, 2, 1), (3, 16, 480, 3, 2, 1), (2, 64, 96, 3, 2, 1)])
def test_stem(hp, cfg):
    N, Fo, size, k, s, p = cfg
    g = torch.Generator().manual_seed(Fo + size)
    x = torch.rand(N, 3, size, size, generator=g)
    w = torch.randn(Fo, 3, k, k, generator=g) * 0.1
    b = torch.randn(Fo, generator=g)
    Ho = (size + 2 * p - k) // s + 1
    ws = torch.empty(hp.stem_ws_bytes(N, 3, Fo, size, size, k, s, p) // 4, device="cuda")
    y = torch.full((N, Fo, Ho, Ho), float("nan"), device="cuda")
    hp.stem_fwd(x.cuda(), w.cuda(), b.cuda(), y, ws, k, s, p)
    close(y, F.conv2d(x, w, b, stride=s, padding=p))
    if hp.stem_x3_supported(3, size, k, s, p):                  # bf16x3 variant, same tolerance
        y3 = torch.full((N, Fo, Ho, Ho), float("nan"), device="cuda")
        hp.stem_fwd(x.cuda(), w.cuda(), b.cuda(), y3, ws, k, s, p, x3=True)
        close(y3, F.conv2d(x, w, b, stride=s, padding=p))
    dy = torch.randn(N, Fo, Ho, Ho, generator=g)
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    F.conv2d(x, wr, br, stride=s, padding=p).backward(dy)
    dW = torch.full((Fo, 3, k, k), float("nan"), device="cuda"); db = torch.full((Fo,), float("nan"), device="cuda")
    hp.stem_wgrad(x.cuda(), dy.cuda(), dW, db, ws, k, s, p)
    close(dW, wr.grad)
    close(db, br.grad)
    if hp.stem_x3_supported(3, size, k, s, p) and size % 16 == 0:
        dW3 = torch.full((Fo, 3, k, k), float("nan"), device="cuda"); db3 = torch.full((Fo,), float("nan"), device="cuda")
        hp.stem_wgrad(x.cuda(), dy.cuda(), dW3, db3, ws, k, s, p, x3=True)
        close(dW3, wr.grad)
        close(db3, br.grad)
    if hp.stem_k3_wgrad_x3_supported(3, Fo, size, size, k, s, p):      # the Resnet stem on the matrix cores (fdet_stem_k3.hip)
        dW3 = torch.full((Fo, 3, k, k), float("nan"), device="cuda"); db3 = torch.full((Fo,), float("nan"), device="cuda")
        hp.stem_wgrad(x.cuda(), dy.cuda(), dW3, db3, ws, k, s, p, x3=True)
        close(dW3, wr.grad)
        close(db3, br.grad)
        dW4 = torch.empty_like(dW3); db4 = torch.empty_like(db3)
        hp.stem_wgrad(x.cuda(), dy.cuda(), dW4, db4, ws, k, s, p, x3=True)
        assert torch.equal(dW3, dW4) and torch.equal(db3, db4)          # fixed-order reduction: bit-reproducible
    if hp.stem_k3_fwd_ps_supported(3, Fo, size, size, k, s, p):         # ... and its forward with a PS (column-strip) output
        from fdet_amd import ps
        yp = ps.PsTensor(N, Fo, Ho, Ho, "cuda")
        ps.stem_fwd_ps(x.cuda(), w.cuda(), b.cuda(), yp, k, s, p)
        close(yp.to_f32(), F.conv2d(x, w, b, stride=s, padding=p))
        before = yp.buf.view(torch.int32).clone()                        # the halo slots it wrote are what the exchange writes
        ps.halo_exchange(yp)
        assert torch.equal(yp.buf.view(torch.int32), before)
        if yp.strips > 1:
            ps.halo_exchange(yp, zero_only=True)
            assert not torch.equal(yp.buf.view(torch.int32), before)


@pytest.mark.parametrize("cfg", [(3, 64, 15, 6, 0), (2, 8, 15, 6, 0), (2, 32, 15, 3, 1), (2, 64, 20, 3, 1), (1, 128, 15, 6, 0)])
def test_head(hp, cfg):
    N, Fi, H, k, p = cfg
    g = torch.Generator().manual_seed(Fi + H + k)
    x = torch.randn(N, Fi, H, H, generator=g)
    w = torch.randn(5, Fi, k, k, generator=g) * 0.05
    b = torch.randn(5, generator=g)
    scale = (torch.rand(N, Fi, generator=g) > 0.5).float() / 0.5
    S = H + 2 * p - k + 1
    y = torch.full((N, 5, S, S), float("nan"), device="cuda")
    hp.head_fwd(x.cuda(), scale.cuda(), w.cuda(), b.cuda(), y, k, p)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = torch.sigmoid(F.conv2d(xr * scale[:, :, None, None], wr, br, padding=p))
    close(y, ref.detach(), 1e-5)
    dy = torch.randn(N, 5, S, S, generator=g)
    ref.backward(dy)
    ws = torch.empty(hp.head_bwd_ws_bytes(N, Fi, H, H, k, p) // 4, device="cuda")
    dx = torch.full((N, Fi, H, H), float("nan"), device="cuda")
    dW = torch.full((5, Fi, k, k), float("nan"), device="cuda"); db = torch.full((5,), float("nan"), device="cuda")
    hp.head_bwd(x.cuda(), scale.cuda(), w.cuda(), y, dy.cuda(), dx, dW, db, ws, k, p)
    close(dx, xr.grad)
    close(dW, wr.grad)
    close(db, br.grad)


@pytest.mark.parametrize("cfg", [(3, 15, True, False), (5, 15, False, False), (2, 12, True, True), (257, 15, True, False)])
def test_head_loss_fused(hp, cfg):
    """fdet_head_loss_fused (training head + yolo_loss + their gradients in one kernel, bf16x3 on the matrix cores) against
    (1) the reference arithmetic on the CPU: Dropout2d scale -> Conv2d(64,5,6) -> sigmoid -> the oracle's yolo_loss per image,
    summed (models/PoolResnet.py:100-102, losses/YoloLoss.py:4-44, models/ModelMeta.py:173-176) and torch autograd, and
    (2) the separate launches it replaces (head_fwd, yolo_loss_fwd_bwd, head_bwd): the loss given y is bit-identical."""
    import oracle as O
    N, H, use_scale, with_nan_targets = cfg
    Fi, k, p = 64, 6, 0
    assert hp.head_loss_fused_supported(Fi, H, H, k, p)
    assert not hp.head_loss_fused_supported(32, H, H, k, p) and not hp.head_loss_fused_supported(Fi, 20, 20, 3, 1)
    S = H - k + 1
    g = torch.Generator().manual_seed(7 * N + H)
    x = torch.randn(N, Fi, H, H, generator=g)
    w = torch.randn(5, Fi, k, k, generator=g) * 0.02
    b = torch.randn(5, generator=g) * 0.1
    scale = ((torch.rand(N, Fi, generator=g) > 0.5).float() / 0.5) if use_scale else None
    boxes = O.synthetic_boxes(N, 48 * S, seed=3)
    gt = torch.stack([O.encode_targets(bb, (48 * S, 48 * S), S) for bb in boxes])
    # ---- CPU reference
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    xs = xr * scale[:, :, None, None] if use_scale else xr
    y_ref = torch.sigmoid(F.conv2d(xs, wr, br))
    l_ref = torch.stack([O.yolo_loss(y_ref[i], gt[i]) for i in range(N)])
    l_ref.sum().backward()
    # ---- fused
    dev = "cuda"
    xc, wc, bc, gc = x.to(dev), w.to(dev), b.to(dev), gt.to(dev)
    sc = scale.to(dev) if use_scale else None
    ws = torch.zeros((hp.head_loss_fused_ws_bytes(N, Fi, H, H, k, p) + 3) // 4, device=dev)
    y = torch.full((N, 5, S, S), float("nan"), device=dev)
    lpi = torch.full((N,), float("nan"), device=dev); lsum = torch.full((1,), float("nan"), device=dev)
    dx = torch.full((N, Fi, H, H), float("nan"), device=dev)
    dW = torch.full((5, Fi, k, k), float("nan"), device=dev); db = torch.full((5,), float("nan"), device=dev)
    for _ in range(2):                                       # twice: the ticket counter in ws must come back to zero
        lsum.fill_(float("nan"))
        hp.head_loss_fused(xc, sc, wc, bc, gc, y, lpi, lsum, dx, dW, db, ws, k, p)
        close(y, y_ref.detach(), 1e-5)
        close(lpi, l_ref.detach(), 1e-4)
        assert abs(float(lsum) - float(l_ref.sum())) <= 1e-4 * max(1.0, float(l_ref.sum()))
        close(dx, xr.grad)
        close(dW, wr.grad)
        close(db, br.grad)
    assert int(ws.view(torch.int32)[-16:].abs().sum()) == 0
    # ---- the launches it replaces: same loss arithmetic on the same y -> identical bits
    lpi2, lsum2, dy2 = hp.yolo_loss_fwd_bwd(y, gc, want_grad=True)
    assert torch.equal(lpi2, lpi) and torch.equal(lsum2, lsum)
    y3 = torch.empty_like(y)
    hp.head_fwd(xc, sc, wc, bc, y3, k, p)
    close(y, y3, 1e-5)
    ws3 = torch.empty(hp.head_bwd_ws_bytes(N, Fi, H, H, k, p) // 4, device=dev)
    dx3 = torch.empty_like(dx); dW3 = torch.empty_like(dW); db3 = torch.empty_like(db)
    hp.head_bwd(xc, sc, wc, y, dy2, dx3, dW3, db3, ws3, k, p)
    close(dx, dx3); close(dW, dW3); close(db, db3)
    if with_nan_targets:
        # a NaN prediction cannot come out of a sigmoid of finite sums; the NaN / inf handling of yolo_loss (Q6) is covered
        # through identical instructions in k_yolo_loss (test_gpu_detect.py); here: an all-zero target image (no object)
        gz = gc.clone(); gz[0] = 0.0
        hp.head_loss_fused(xc, sc, wc, bc, gz, y, lpi, lsum, dx, dW, db, ws, k, p)
        lz = O.yolo_loss(y_ref[0].detach(), gt[0] * 0.0)
        assert abs(float(lpi[0]) - float(lz)) <= 1e-4 * max(1.0, float(lz))


@pytest.mark.parametrize("cfg", [(3, 15, 15, 3, True), (2, 10, 10, 2, False), (5, 15, 15, 8, True), (1, 7, 9, 1, True)])
def test_block_chain_fwd_bwd(hp, cfg):
    """LDS-resident residual-block chain (fdet_block_chain_{fwd,bwd}_bf16x3) against torch CPU fp32:
    forward a/c/out of every block, backward dz1/dz2 of every block and the input gradient.
    Reference semantics: models/PoolResnet.py:33-43 with pool == 1 (conv-lrelu-conv-lrelu-dropout2d-skip)."""
    N, H, W, nb, use_scale = cfg
    C = 64
    assert hp.block_chain_supported(C, H, W)
    g = torch.Generator().manual_seed(N * 100 + H + nb)
    x = torch.randn(N, C, H, W, generator=g)
    Ws = [(torch.randn(C, C, 3, 3, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1,
           torch.randn(C, C, 3, 3, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1) for _ in range(nb)]
    scales = [((torch.rand(N, C, generator=g) > 0.25).float() / 0.75) for _ in range(nb)] if use_scale else None
    # ---- torch reference
    xr = x.clone().requires_grad_(True)
    h = xr
    a_ref, c_ref, o_ref, z1, z2 = [], [], [], [], []
    for k in range(nb):
        w1, b1, w2, b2 = Ws[k]
        p1 = F.conv2d(h, w1, b1, padding=1); p1.retain_grad(); z1.append(p1)
        a = F.leaky_relu(p1, 0.2)
        p2 = F.conv2d(a, w2, b2, padding=1); p2.retain_grad(); z2.append(p2)
        c = F.leaky_relu(p2, 0.2)
        h = (c * scales[k][:, :, None, None] if use_scale else c) + h
        a_ref.append(a); c_ref.append(c); o_ref.append(h)
    dout = torch.randn(N, C, H, W, generator=g)
    h.backward(dout)
    # ---- device
    nf, nbk = hp.packed_sizes(C, C)
    wf1, wb1, wf2, wb2 = [], [], [], []
    for (w1, b1, w2, b2) in Ws:
        for w, lf, lb in ((w1, wf1, wb1), (w2, wf2, wb2)):
            f_ = torch.empty(nf, device="cuda"); b_ = torch.empty(nbk, device="cuda")
            hp.pack_conv3x3_weights(w.cuda(), f_, b_, x3=True)
            lf.append(f_); lb.append(b_)
    mk = lambda: [torch.full((N, C, H, W), float("nan"), device="cuda") for _ in range(nb)]
    a_d, c_d, o_d = mk(), mk(), mk()
    sc_d = [s_.cuda() for s_ in scales] if use_scale else None
    hp.block_chain_fwd(x.cuda(), wf1, [w[1].cuda() for w in Ws], wf2, [w[3].cuda() for w in Ws], sc_d, a_d, c_d, o_d)
    for k in range(nb):
        close(a_d[k], a_ref[k].detach())
        close(c_d[k], c_ref[k].detach())
        close(o_d[k], o_ref[k].detach())
    # eval flavour: nothing kept, only the last output
    last = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.block_chain_fwd(x.cuda(), wf1, [w[1].cuda() for w in Ws], wf2, [w[3].cuda() for w in Ws], sc_d, None, None,
                       [None] * (nb - 1) + [last])
    assert torch.equal(last, o_d[-1])
    dz1_d, dz2_d = mk(), mk()
    dx_d = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.block_chain_bwd(dout.cuda(), wb1, wb2, sc_d, [t.detach().cuda() for t in a_ref], [t.detach().cuda() for t in c_ref],
                       dz1_d, dz2_d, dx_d)
    for k in range(nb):
        close(dz1_d[k], z1[k].grad)
        close(dz2_d[k], z2[k].grad)
    close(dx_d, xr.grad)


POOL_SHAPES = [(3, 64, 60, 60), (4, 64, 30, 30), (2, 32, 30, 30), (5, 64, 28, 30), (1, 64, 6, 58), (2, 128, 30, 30),
               (33, 64, 60, 60), (1, 32, 2, 62), (2, 32, 4, 10)]


@pytest.mark.parametrize("shape", POOL_SHAPES)
def test_pooled_block_fused_into_conv_epilogues(hp, shape):
    """Pooled residual-block tail inside the conv kernels (fdet_conv3x3_fwd_pool_bf16x3 / fdet_pool_route_bwd /
    fdet_conv3x3_dgrad_unpool_bf16x3) against torch fp32 CPU of models/PoolResnet.py:36-42 and its autograd.
    Values: 1e-4 of the tensor scale.  Routing: the pooled VALUE and its gradient are what is compared, so a window
    whose two largest entries differ by less than the conv's rounding may legitimately route to the other one --
    such windows are excluded where the runner-up is within 1e-4 (counted, must stay rare)."""
    N, C, H, W = shape
    assert hp.pool_fusion_supported(C, C, H, W)
    g = torch.Generator().manual_seed(N * 7 + C + H + W)
    a = torch.randn(N, C, H, W, generator=g)
    w2 = torch.randn(C, C, 3, 3, generator=g) * 0.1
    b2 = torch.randn(C, generator=g)
    skip = torch.randn(N, C, H, W, generator=g)
    scale = (torch.rand(N, C, generator=g) > 0.25).float() / 0.75
    nf, nb = hp.packed_sizes(C, C)
    wf = torch.empty(nf, device="cuda"); wb = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w2.cuda(), wf, wb, x3=True)
    out = torch.full((N, C, H // 2, W // 2), float("nan"), device="cuda")
    route = torch.full((N, C, H // 2, W // 2), 255, dtype=torch.uint8, device="cuda")
    hp.conv3x3_fwd_pool(a.cuda(), wf, b2.cuda(), skip.cuda(), scale.cuda(), out, route)
    c = F.leaky_relu(F.conv2d(a, w2, b2, padding=1), 0.2)
    e = c * scale[:, :, None, None] + skip
    ref, ref_idx = F.max_pool2d(e, 2, return_indices=True)
    close(out, ref)
    # eval flavour: no routing bytes, no dropout
    out_e = torch.full_like(out, float("nan"))
    hp.conv3x3_fwd_pool(a.cuda(), wf, b2.cuda(), skip.cuda(), None, out_e, None)
    close(out_e, F.max_pool2d(c + skip, 2))
    # routing bytes: argmax (ATen scan order) and sign bits of c, wherever the decision is not within rounding
    r = route.cpu().int()
    arg = (r >> 4) & 3
    wy = torch.arange(H // 2).view(1, 1, -1, 1) * 2
    wx = torch.arange(W // 2).view(1, 1, 1, -1) * 2
    ref_arg = ((ref_idx // W) - wy) * 2 + ((ref_idx % W) - wx)
    ew = e.unfold(2, 2, 2).unfold(3, 2, 2).reshape(N, C, H // 2, W // 2, 4)
    top2 = ew.topk(2, dim=-1).values
    decided = (top2[..., 0] - top2[..., 1]) > 1e-4 * max(1.0, float(e.abs().max()))
    assert float(decided.float().mean()) > 0.99
    assert torch.equal(arg[decided], ref_arg[decided].int())
    cw = c.unfold(2, 2, 2).unfold(3, 2, 2).reshape(N, C, H // 2, W // 2, 4)
    for k in range(4):
        clear = cw[..., k].abs() > 1e-4 * max(1.0, float(c.abs().max()))
        assert torch.equal(((r >> k) & 1)[clear].bool(), (cw[..., k] > 0)[clear])
    assert int((r >> 6).max()) == 0
    # backward: dz2 = unpool(dout)*scale*lrelu'(c) from the bytes alone
    dout = torch.randn(N, C, H // 2, W // 2, generator=g)
    dz2 = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.pool_route_bwd(dout.cuda(), route, scale.cuda(), dz2)
    # reference built from the KERNEL's routing (the decision itself was checked above)
    de_ref = torch.zeros(N, C, H // 2, W // 2, 4)
    de_ref.scatter_(-1, arg.long().unsqueeze(-1), dout.unsqueeze(-1))
    de_full = de_ref.reshape(N, C, H // 2, W // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H, W)
    bits = torch.stack([(r >> k) & 1 for k in range(4)], -1).reshape(N, C, H // 2, W // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H, W)
    dz2_ref = de_full * scale[:, :, None, None] * torch.where(bits > 0, 1.0, 0.2)
    assert torch.equal(dz2.cpu(), dz2_ref)                     # pure routing: bit-exact
    # ... and the skip-path gradient added inside conv1's data gradient
    w1 = torch.randn(C, C, 3, 3, generator=g) * 0.1
    wf1 = torch.empty(nf, device="cuda"); wb1 = torch.empty(nb, device="cuda")
    hp.pack_conv3x3_weights(w1.cuda(), wf1, wb1, x3=True)
    dz1 = torch.randn(N, C, H, W, generator=g)
    dx = torch.full((N, C, H, W), float("nan"), device="cuda")
    hp.conv3x3_dgrad_unpool(dz1.cuda(), wb1, C, dout.cuda(), route, dx)
    close(dx, F.conv_transpose2d(dz1, w1, padding=1) + de_full)
    # end to end against autograd of the reference block tail (decided windows only)
    er = e.clone().requires_grad_(True)
    F.max_pool2d(er, 2).backward(dout)
    full_decided = decided.reshape(N, C, H // 2, W // 2, 1, 1).expand(-1, -1, -1, -1, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(N, C, H, W)
    assert torch.equal(de_full[full_decided], er.grad[full_decided])


PW_SHAPES = [(2, 16, 32, 24, 24), (3, 32, 64, 60, 60), (2, 64, 128, 15, 15), (2, 128, 256, 30, 30), (2, 256, 5, 15, 15),
             (3, 256, 5, 7, 7), (1, 128, 5, 60, 60), (2, 24, 40, 9, 13), (1, 3, 7, 5, 6),
             (1, 160, 288, 6, 6), (2, 288, 96, 5, 7)]      # (the last two: beyond the one-workgroup-per-slab weight gradient)


@pytest.mark.parametrize("shape", PW_SHAPES)
def test_pointwise_gemm_fwd_dgrad_wgrad(hp, shape):
    """1x1 conv / per-position Linear as dense GEMMs (fdet_pointwise_*_bf16x3) against torch fp32 CPU: forward with bias
    and optional LeakyReLU, data gradient with the fused add, weight and bias gradient.  1e-4 of the tensor scale.
    Shapes: the SSD skip convs and heads (models/SSD.py:24-30,183-185), odd channel counts, planes that are not multiples
    of 4 / 16 positions (unaligned rows: the dword path)."""
    N, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(N + Ci + Co + H)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    wf, wb = hp.pointwise_pack(w.cuda())
    y = torch.full((N, Co, H, W), float("nan"), device="cuda")
    hp.pointwise_fwd(x.cuda(), wf, b.cuda(), y)
    ref = F.conv2d(x, w, b)
    close(y, ref)
    y2 = torch.full((N, Co, H, W), float("nan"), device="cuda")
    hp.pointwise_fwd(x.cuda(), wf, None, y2, slope=0.2)
    close(y2, F.leaky_relu(F.conv2d(x, w), 0.2))
    dz = torch.randn(N, Co, H, W, generator=g)
    add = torch.randn(N, Ci, H, W, generator=g)
    dx = torch.full((N, Ci, H, W), float("nan"), device="cuda")
    hp.pointwise_dgrad(dz.cuda(), wb, dx, add=add.cuda())
    close(dx, F.conv_transpose2d(dz, w) + add)
    hp.pointwise_dgrad(dz.cuda(), wb, dx)
    close(dx, F.conv_transpose2d(dz, w))
    dW = torch.full((Co, Ci, 1, 1), float("nan"), device="cuda"); db = torch.full((Co,), float("nan"), device="cuda")
    hp.pointwise_wgrad(x.cuda(), dz.cuda(), dW, db)
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    F.conv2d(x, wr, br).backward(dz)
    close(dW, wr.grad)
    close(db, br.grad)
