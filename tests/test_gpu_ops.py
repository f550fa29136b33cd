"""GPU parity, op level: each HIP kernel family against the oracle through the C ABI.
Tolerances: a single fp32 GEMM/conv vs the float64 oracle -> 2e-5 abs on O(1) values."""
import ctypes as C

import numpy as np
import pytest

import taco_oracle as O
from util import tiny_hp, build_model, dev, ptr, stream, maxabs

pytestmark = pytest.mark.gpu
TOL = 3e-5
TOL_SPLIT = 1e-4   # feed-forward GEMMs run as 3-term split-bf16 products (~2^-16 relative per product): N(0,1) inputs, K = 3 x 128 channels


@pytest.fixture(scope="module")
def ctx():
    import taco_amd
    ohp = tiny_hp()
    w = O.init_weights(ohp, 1, 7)
    m = build_model(ohp, w)
    return ohp, w, m, taco_amd._lib


def _conv(ctx, layer, x, act, mpw=1):
    import torch
    ohp, w, m, L = ctx
    B, T, _ = x.shape
    cout = w[layer + "/kernel"].shape[-1]
    out = torch.full((B, T, cout), float("nan"), device="cuda")
    xd = dev(x, torch.float32)      # keep a reference: the allocator may reuse a freed temporary
    L.check(m._lib.taco_conv1d_bn_f32(m._handle, stream(), layer.encode(), ptr(xd), B, T, act, mpw, ptr(out)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3])
def test_conv_bank_member(ctx, k, cfg):
    ohp, w, m, L = ctx
    rs = np.random.RandomState(100 + k)
    x = rs.randn(3, 37, ohp.enc_prenet_sizes[-1])
    name = "encoder_cbhg/conv_bank/conv1d_%d" % k
    m._lib.taco_debug_force_gemm_config(m._handle, cfg)
    try:
        y = _conv(ctx, name, x, 1)
    finally:
        m._lib.taco_debug_force_gemm_config(m._handle, -1)
    assert maxabs(y, O.conv1d_bn(x, w, name, O.relu)) < TOL


def test_conv_rows_cross_tile_and_batch_boundaries(ctx):
    # T not a multiple of any tile, B*T > 128 so several M-tiles, halo must not leak across batch rows
    ohp, w, m, L = ctx
    rs = np.random.RandomState(5)
    x = rs.randn(5, 61, ohp.enc_prenet_sizes[-1])
    for k in (4, 5):
        name = "encoder_cbhg/conv_bank/conv1d_%d" % k
        assert maxabs(_conv(ctx, name, x, 1), O.conv1d_bn(x, w, name, O.relu)) < TOL


def test_projection_with_fused_maxpool(ctx):
    ohp, w, m, L = ctx
    rs = np.random.RandomState(6)
    C_in = ohp.enc_bank_size * ohp.enc_bank_channel_size      # > 64 channels -> several LDS chunks
    x = rs.randn(2, 29, C_in)
    ref = O.conv1d_bn(O.maxpool_same_stride1(x, 2), w, "encoder_cbhg/proj_1", O.relu)
    assert maxabs(_conv(ctx, "encoder_cbhg/proj_1", x, 1, mpw=2), ref) < TOL_SPLIT
    ref = O.conv1d_bn(x, w, "encoder_cbhg/proj_1", None)
    assert maxabs(_conv(ctx, "encoder_cbhg/proj_1", x, 0, mpw=1), ref) < TOL_SPLIT


@pytest.mark.parametrize("layer,rows", [("prenet/dense_2", 50), ("linear", 70), ("post_cbhg/dense", 33),
                                        ("attention/memory_layer", 19), ("decoder/prenet/dense_2", 3),
                                        ("decoder/frame_projection", 9), ("attention/query_layer", 33),
                                        ("decoder/concat_projection", 64)])
def test_dense_layers_big_and_skinny(ctx, layer, rows):
    import torch
    ohp, w, m, L = ctx
    k = w[layer + "/kernel"]
    rs = np.random.RandomState(rows)
    x = rs.randn(rows, k.shape[0])
    out = torch.full((rows, k.shape[1]), float("nan"), device="cuda")
    xd = dev(x, torch.float32)
    L.check(m._lib.taco_dense_f32(m._handle, stream(), layer.encode(), ptr(xd), rows, 1, ptr(out)))
    torch.cuda.synchronize()
    ref = O.dense(x, w, layer, O.relu, bias=(layer + "/bias") in w)
    assert maxabs(out.cpu().numpy(), ref) < TOL


def test_highway(ctx):
    import torch
    ohp, w, m, L = ctx
    rs = np.random.RandomState(8)
    for scope, D in (("encoder_cbhg", ohp.enc_rnn_size), ("post_cbhg", ohp.post_rnn_size)):
        x = rs.randn(77, D)
        out = torch.full((77, D), float("nan"), device="cuda")
        xd = dev(x, torch.float32)
        L.check(m._lib.taco_highway_f32(m._handle, stream(), (scope + "/highway_2").encode(), ptr(xd), 77, ptr(out)))
        torch.cuda.synchronize()
        assert maxabs(out.cpu().numpy(), O.highwaynet(x, w, scope + "/highway_2")) < TOL


def test_stop_steps_per_group_of_rows(ctx):
    """taco_stop_steps: helpers.py:29 + dynamic_decode on a finished mel buffer, per group of rows (requests served together):
    stop = min(max over the group's rows of (first all-zero step) + 1, n); rows that never emit an all-zero step keep it at n."""
    import torch
    ohp, w, m, L = ctx
    rs = np.random.RandomState(12)
    B, n, width, rows = 12, 37, 20, 3
    y = rs.randn(B, n, width).astype(np.float32)
    first = rs.randint(0, n + 6, size=B)                  # >= n: the row never stops
    for b in range(B):
        if first[b] < n:
            y[b, first[b]] = 0.0
            if first[b] + 3 < n:
                y[b, first[b] + 3] = 0.0                  # later zero steps do not matter
        y[b, :min(first[b], n), 0] += 10.0                # no accidental earlier zero step
    want = np.array([min(max(min(f, n) for f in first[g * rows:(g + 1) * rows]) + 1, n) for g in range(B // rows)], np.int32)
    yd = dev(y, torch.float32)
    out = torch.full((B // rows,), -1, dtype=torch.int32, device="cuda")
    L.check(m._lib.taco_stop_steps(stream(), ptr(yd), B, n, width, rows, ptr(out)))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want), (out.cpu().numpy(), want, first)
    with pytest.raises(L.TacoError):
        L.check(m._lib.taco_stop_steps(stream(), ptr(yd), B, n, width, 5, ptr(out)))      # 12 rows do not split into groups of 5


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7, 9, 10])
def test_split_bf16_gemm_every_tile_shape(ctx, tile):
    """k_gemm_bf3 under each of its tile shapes (1: 128x64, 2: 128x128, 3: 64x256, 4: 64x64, 5: 64x64 with four wave groups
    splitting K inside the workgroup, 7: 64x256 by 1x8 waves, 9: 64x128 by 1x4 waves, 10: tile 7 with two wave groups splitting K): conv with taps over ragged row counts and several batch rows, projection with the fused
    max-pool and several LDS chunks, dense with an odd column count, highway (two weight matrices).  The automatic choice picks
    4/5 at these sizes, so the large-layer tiles (7, 9 and the older 1-3) are pinned here."""
    import torch
    ohp, w, m, L = ctx
    L.check(m._lib.taco_debug_set_bf3(m._handle, 1, tile))
    try:
        rs = np.random.RandomState(40 + tile)
        x = rs.randn(5, 61, ohp.enc_prenet_sizes[-1])
        for k in (1, 4, 5):
            name = "encoder_cbhg/conv_bank/conv1d_%d" % k
            assert maxabs(_conv(ctx, name, x, 1), O.conv1d_bn(x, w, name, O.relu)) < TOL_SPLIT
        xc = rs.randn(3, 70, ohp.enc_bank_size * ohp.enc_bank_channel_size)
        ref = O.conv1d_bn(O.maxpool_same_stride1(xc, 2), w, "encoder_cbhg/proj_1", O.relu)
        assert maxabs(_conv(ctx, "encoder_cbhg/proj_1", xc, 1, mpw=2), ref) < TOL_SPLIT
        xp = rs.randn(2, 131, ohp.post_bank_size * ohp.post_bank_channel_size)
        ref = O.conv1d_bn(xp, w, "post_cbhg/proj_1", O.relu)
        assert maxabs(_conv(ctx, "post_cbhg/proj_1", xp, 1, mpw=1), ref) < TOL_SPLIT
        rows = 201
        xd0 = rs.randn(rows, w["linear/kernel"].shape[0])
        out = torch.full((rows, w["linear/kernel"].shape[1]), float("nan"), device="cuda")
        xd = dev(xd0, torch.float32)
        L.check(m._lib.taco_dense_f32(m._handle, stream(), b"linear", ptr(xd), rows, 0, ptr(out)))
        torch.cuda.synchronize()
        assert maxabs(out.cpu().numpy(), O.dense(xd0, w, "linear", None)) < TOL_SPLIT
        for scope, D in (("encoder_cbhg", ohp.enc_rnn_size), ("post_cbhg", ohp.post_rnn_size)):
            xh = rs.randn(150, D)
            outh = torch.full((150, D), float("nan"), device="cuda")
            xhd = dev(xh, torch.float32)
            L.check(m._lib.taco_highway_f32(m._handle, stream(), (scope + "/highway_1").encode(), ptr(xhd), 150, ptr(outh)))
            torch.cuda.synchronize()
            assert maxabs(outh.cpu().numpy(), O.highwaynet(xh, w, scope + "/highway_1")) < TOL_SPLIT
    finally:
        L.check(m._lib.taco_debug_set_bf3(m._handle, 1, 0))


@pytest.mark.parametrize("persist", [1, 0])
@pytest.mark.parametrize("B", [1, 5, 17, 32, 33])
def test_bigru_with_lengths_and_init_state(ctx, B, persist):
    """persist=1: one persistent row-parallel launch (weights streamed from L2 every step);
    persist=0: two launches per time step."""
    import torch
    ohp, w, m, L = ctx
    m._lib.taco_debug_set_persistent(m._handle, persist)
    rs = np.random.RandomState(B)
    T, H = 12, ohp.enc_rnn_size
    x = rs.randn(B, T, H)
    lens = rs.randint(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    init = rs.randn(B, 2 * H) * 0.5
    ref = O.bidirectional_gru(x, lens, w, "encoder_cbhg/bigru", init)
    out = torch.full((B, T, 2 * H), float("nan"), device="cuda")
    n = int(m._lib.taco_stage_workspace_bytes(m._handle, B, T))
    ws = torch.empty((n,), dtype=torch.uint8, device="cuda")
    xd, ld, idv = dev(x, torch.float32), dev(lens), dev(init, torch.float32)
    L.check(m._lib.taco_bigru_f32(m._handle, stream(), b"encoder_cbhg", ptr(xd), ptr(ld), ptr(idv), B, T, ptr(out), ptr(ws), n))
    torch.cuda.synchronize()
    assert maxabs(out.cpu().numpy(), ref) < 1e-4
    # post-net flavour: no lengths, zero init
    Hp = ohp.post_rnn_size
    xp = rs.randn(B, T, Hp)
    outp = torch.full((B, T, 2 * Hp), float("nan"), device="cuda")
    xpd = dev(xp, torch.float32)
    L.check(m._lib.taco_bigru_f32(m._handle, stream(), b"post_cbhg", ptr(xpd), ptr(None), ptr(None), B, T, ptr(outp), ptr(ws), n))
    torch.cuda.synchronize()
    assert maxabs(outp.cpu().numpy(), O.bidirectional_gru(xp, None, w, "post_cbhg/bigru")) < 1e-4
    m._lib.taco_debug_set_persistent(m._handle, 1)
    m.check_device_errors()


def test_bigru_wave_local_scan_is_bit_identical_to_the_four_barrier_one():
    """k_bigru_resw (persist 7; the default before k_bigru_xcd: gate and state exchanges kept inside the wave that owns the K-slice, two barriers per
    step, column-permuted weight packs read with 16-byte loads) performs the arithmetic of its predecessor k_bigru_resu (persist 6,
    four barriers) in the same order: identical bits, with ragged lengths and an initial state too (A.7 masking, modules.py:82-86)."""
    import torch
    import taco_amd
    ohp = O.OracleHParams(max_iters=4)
    w = O.init_weights(ohp, 1, 13)
    m = build_model(ohp, w)
    rs = np.random.RandomState(14)
    B, T, H = 7, 50, ohp.post_rnn_size
    x = rs.randn(B, T, H) * 0.5
    lens = rs.randint(0, T + 1, size=B).astype(np.int32); lens[0] = T; lens[1] = 0
    init = rs.randn(B, 2 * H) * 0.5
    xd, ld, idv = dev(x, torch.float32), dev(lens), dev(init, torch.float32)
    n = int(m._lib.taco_stage_workspace_bytes(m._handle, B, T))
    ws = torch.empty((n,), dtype=torch.uint8, device="cuda")
    got = {}
    for persist in (7, 6):
        m._lib.taco_debug_set_persistent(m._handle, persist)
        for tag, (lp, ip) in (("plain", (ptr(None), ptr(None))), ("ragged", (ptr(ld), ptr(idv)))):
            out = torch.full((B, T, 2 * H), float("nan"), device="cuda")
            taco_amd._lib.check(m._lib.taco_bigru_f32(m._handle, stream(), b"post_cbhg", ptr(xd), lp, ip, B, T, ptr(out), ptr(ws), n))
            torch.cuda.synchronize()
            got[(persist, tag)] = out.cpu().numpy()
    m._lib.taco_debug_set_persistent(m._handle, 1)
    m.check_device_errors()
    for tag in ("plain", "ragged"):
        assert np.array_equal(got[(7, tag)], got[(6, tag)]), tag
    assert maxabs(got[(6, "plain")], O.bidirectional_gru(x, None, w, "post_cbhg/bigru")) < 1e-4
    assert maxabs(got[(6, "ragged")], O.bidirectional_gru(x, lens, w, "post_cbhg/bigru", init)) < 1e-4


@pytest.mark.parametrize("persist", [1, 2, 3, 4, 5, 6, 8, 9])
@pytest.mark.parametrize("B", [32, 5])
def test_bigru_persistent_full_width_repeatable(persist, B):
    """Full-width BiGRUs, T=64, three runs: results must match the oracle and be bit-identical run to run.
    persist=1: k_bigru_res (recurrent weights resident in registers + LDS, one row per workgroup);
    persist=2: k_bigru_rows (weights re-streamed from L2 every step)."""
    import torch
    import taco_amd
    ohp = O.OracleHParams(max_iters=4)
    w = O.init_weights(ohp, 1, 11)
    m = build_model(ohp, w)
    m._lib.taco_debug_set_persistent(m._handle, persist)
    rs = np.random.RandomState(12)
    T, H = 64, ohp.post_rnn_size
    x = rs.randn(B, T, H) * 0.5
    ref = O.bidirectional_gru(x, None, w, "post_cbhg/bigru")
    xd = dev(x, torch.float32)
    n = int(m._lib.taco_stage_workspace_bytes(m._handle, B, T))
    ws = torch.empty((n,), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(3):
        out = torch.full((B, T, 2 * H), float("nan"), device="cuda")
        taco_amd._lib.check(m._lib.taco_bigru_f32(m._handle, stream(), b"post_cbhg", ptr(xd), ptr(None), ptr(None), B, T,
                                                  ptr(out), ptr(ws), n))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    m.check_device_errors()
    assert maxabs(outs[0], ref) < 1e-4
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    # encoder flavour with ragged lengths and an initial state (deepvoice encoder_rnn_init, modules.py:82-86)
    He = ohp.enc_rnn_size
    xe = rs.randn(B, T, He) * 0.5
    lens = rs.randint(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    init = rs.randn(B, 2 * He) * 0.5
    xed, ld, idv = dev(xe, torch.float32), dev(lens), dev(init, torch.float32)
    oute = torch.full((B, T, 2 * He), float("nan"), device="cuda")
    taco_amd._lib.check(m._lib.taco_bigru_f32(m._handle, stream(), b"encoder_cbhg", ptr(xed), ptr(ld), ptr(idv), B, T,
                                              ptr(oute), ptr(ws), n))
    torch.cuda.synchronize()
    m.check_device_errors()
    assert maxabs(oute.cpu().numpy(), O.bidirectional_gru(xe, lens, w, "encoder_cbhg/bigru", init)) < 1e-4


@pytest.mark.parametrize("name,res", [("decoder/attention_gru", False), ("decoder/gru_1", True), ("decoder/gru_2", True)])
def test_decoder_gru_cell(ctx, name, res):
    import torch
    ohp, w, m, L = ctx
    rs = np.random.RandomState(3)
    I = w[name + "/gates/kernel"].shape[0] - w[name + "/candidate/bias"].shape[0]
    H = w[name + "/candidate/bias"].shape[0]
    R = 6
    x, h = rs.randn(R, I), rs.randn(R, H)
    hn = O.gru_cell(x, h, w, name)
    hd = dev(h, torch.float32)
    outr = torch.full((R, H), float("nan"), device="cuda") if res else None
    ws = torch.empty((1 << 16,), dtype=torch.uint8, device="cuda")
    xd = dev(x, torch.float32)
    L.check(m._lib.taco_gru_cell_f32(m._handle, stream(), name.encode(), ptr(xd), ptr(hd), R, ptr(outr), ptr(ws), 1 << 16))
    torch.cuda.synchronize()
    assert maxabs(hd.cpu().numpy(), hn) < TOL
    if res:
        assert maxabs(outr.cpu().numpy(), hn + x) < TOL


@pytest.mark.parametrize("atype", ["bah", "bah_norm", "bah_mon"])
@pytest.mark.parametrize("T_in", [1, 7, 64, 150])
def test_attention_step(atype, T_in):
    import torch
    import taco_amd
    ohp = tiny_hp(attention_type=atype)
    w = O.init_weights(ohp, 1, 9)
    if atype == "bah_mon":
        w["attention/attention_score_bias"] = np.float32(-0.7).reshape(())
    m = build_model(ohp, w)
    rs = np.random.RandomState(T_in)
    B, A, D = 4, ohp.attention_size, 2 * ohp.enc_rnn_size
    cell = rs.randn(B, ohp.attention_state_size)
    keys, values = rs.randn(B, T_in, A), rs.randn(B, T_in, D)
    prev = rs.dirichlet(np.ones(T_in), B)
    q = O.dense(cell, w, "attention/query_layer", bias=False)
    a_ref = O.attention_alignments(q, keys, prev, w, atype)
    c_ref = np.einsum("bj,bjd->bd", a_ref, values)
    al = torch.full((B, T_in), float("nan"), device="cuda")
    cx = torch.full((B, D), float("nan"), device="cuda")
    ws = torch.empty((1 << 16,), dtype=torch.uint8, device="cuda")
    cd, kd, vd, pd = (dev(t, torch.float32) for t in (cell, keys, values, prev))
    taco_amd._lib.check(m._lib.taco_attention_step_f32(
        m._handle, stream(), ptr(cd), ptr(kd), ptr(vd), ptr(pd), B, T_in, ptr(al), ptr(cx), ptr(ws), 1 << 16))
    torch.cuda.synchronize()
    assert maxabs(al.cpu().numpy(), a_ref) < 2e-6
    assert maxabs(cx.cpu().numpy(), c_ref) < 2e-5


@pytest.mark.parametrize("prioritize", [False, True])
def test_add_loss_kernel(prioritize):
    """tacotron.py:274-302 incl. the prioritize_loss band (165 Hz .. 5 kHz of num_freq bins)."""
    import torch
    import taco_amd
    rs = np.random.RandomState(5)
    B, T, M, F = 3, 37, 80, 1025
    mo, mt = rs.rand(B, T, M), rs.rand(B, T, M)
    lo, lt = rs.rand(B, T, F), rs.rand(B, T, F)
    coeff = np.array([1.0, 0.5, 2.0])
    ref = O.add_loss(mo, mt, lo, lt, coeff, prioritize_loss=prioritize, sample_rate=24000)
    out = taco_amd.train_ops.l1_losses(dev(mo, torch.float32), dev(mt, torch.float32), dev(lo, torch.float32), dev(lt, torch.float32),
                                       dev(coeff, torch.float32), prioritize_loss=prioritize, sample_rate=24000).cpu().numpy()
    want = [ref["loss"], ref["mel_loss"], ref["linear_loss"], ref["loss_without_coeff"]]
    assert np.allclose(out, want, rtol=2e-6, atol=1e-7), (out, want)


@pytest.mark.parametrize("mode,rnd", [(0, True), (0, False), (1, True)])
def test_flat_adam_clip_lr_schedule(mode, rnd):
    """tacotron.py:305-336: LR schedule, clip_by_global_norm(1.0), TF-form Adam over a flat buffer; 5 updates."""
    import torch
    import taco_amd
    rs = np.random.RandomState(6)
    n = 100003
    p0 = rs.randn(n).astype(np.float32)
    opt = taco_amd.train_ops.FlatAdam(dev(p0.copy()), decay_learning_rate_mode=mode, is_randomly_initialized=rnd)
    p, m, v = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 6):
        g = (rs.randn(n) * (3.0 if t % 2 else 0.001)).astype(np.float32)          # norm above and below the clip
        lr = O.learning_rate(t - 1, 0.002, mode, rnd)
        assert abs(opt.learning_rate - lr) < 1e-9 + 1e-6 * lr
        p, m, v, gn = O.adam_clip_step(p, g.astype(np.float64), m, v, t, lr)
        opt.step(dev(g))
        torch.cuda.synchronize()
        assert abs(float(opt.gnorm.item()) - gn) < 1e-4 * gn
    assert np.abs(opt.params.cpu().numpy() - p).max() < 2e-6
    assert np.abs(opt.m.cpu().numpy() - m).max() < 1e-6 and np.abs(opt.v.cpu().numpy() - v).max() < 1e-6


def test_attention_trim_matches_reference_walk():
    """synthesizer.py:242-262 (attention_trim && end_of_sentence) as a device kernel vs its line-by-line restatement."""
    import ctypes as C
    import torch
    import taco_amd
    lib = taco_amd._lib.load_library()
    rs = np.random.RandomState(3)
    for case in range(6):
        N, T_in, n, r = 5, 11 + case, 23 + 7 * case, 2 + case % 4
        al = rs.rand(N, T_in, n).astype(np.float32) * 0.1
        seq_len = rs.randint(3, T_in + 1, size=N).astype(np.int32)
        for b in range(N):              # a mostly monotonic path, with dwell on the last symbol for some rows, ties and early stops for others
            pos = np.minimum((np.arange(n) * (T_in + 3) // n), T_in - 1)
            if b % 2:
                pos = np.minimum(pos, seq_len[b] - 1)
            if b == 3:
                pos[:] = 0
            al[b, pos, np.arange(n)] += 1.0
        want = np.array([O.attention_trim_end(al[b], int(seq_len[b]), r) for b in range(N)])
        ad, sd = dev(al), dev(seq_len)
        out = torch.zeros(N, dtype=torch.int32, device="cuda")
        taco_amd._lib.check(lib.taco_attention_trim(stream(), ptr(ad), ptr(sd), N, T_in, n, r, ptr(out)))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), (case, out.cpu().numpy(), want)


def test_attention_trim_kernel_equals_what_the_reference_kept():
    """The device kernel against the REFERENCE itself: the 120 alignments on which the reference's own plot_graph_and_save_audio ran
    (tests/golden/trim_vectors.npz, written by tools/make_reference_vectors.py) -> the same number of kept frames, every one."""
    import os
    import torch
    import taco_amd
    lib = taco_amd._lib.load_library()
    tv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trim_vectors.npz"))
    r = int(tv["reduction_factor"][0])
    for al, (T_in, n), L, want in zip(tv["alignments"], tv["dims"], tv["sequence_len"], tv["spec_end_idx"]):
        ad = dev(np.ascontiguousarray(al[:T_in, :n][None].astype(np.float32)))
        sd = dev(np.array([L], np.int32))
        out = torch.zeros(1, dtype=torch.int32, device="cuda")
        taco_amd._lib.check(lib.taco_attention_trim(stream(), ptr(ad), ptr(sd), 1, int(T_in), int(n), r, ptr(out)))
        assert int(out.item()) == int(want), (T_in, n, L, int(out.item()), int(want))
