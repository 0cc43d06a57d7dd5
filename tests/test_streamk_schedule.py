"""The even ("stream-K") schedule of the fp32 NT GEMM (csrc/xv_gemm.hip, xv_gemm_nt_sk_kernel / xv_launch_gemm_nt) restated in Python and
checked exhaustively on the layer shapes: the kernel's integer formulas must tile the (tile, K-step) units exactly once, agree on who
shares a tile, and never hand two shares of one workgroup the same slab slot.  (Host logic only; the kernel itself is compared with the
oracle in tests/test_gpu_ops.py.)"""
import pytest


def schedule(tiles, nk, P):
    total = tiles * nk
    begin = [w * total // P for w in range(P + 1)]                       # u = w * total / P
    owner = lambda u: ((u + 1) * P - 1) // total                         # ntsk_owner
    shares = []                                                          # (worker, tile, kt0, kt1, slot)
    for w in range(P):
        u, u_end = begin[w], begin[w + 1]
        first_tile = u // nk
        while u < u_end:
            tile = u // nk
            kt0 = u - tile * nk
            kt1 = min(nk, kt0 + (u_end - u))
            shares.append((w, tile, kt0, kt1, 0 if tile == first_tile else 1))
            u += kt1 - kt0
    return total, begin, owner, shares


SHAPES = [  # (rows, cols, K) -> tiles = ceil(rows/128) * ceil(cols/128), nk = ceil(K/16)
    (24576, 512, 2560), (25088, 512, 2560), (23808, 512, 3584), (24576, 512, 3584), (23808, 512, 512), (23808, 1500, 512), (23808, 512, 1500),
    (18688, 512, 2560), (18304, 512, 3584), (12544, 512, 160), (128, 7352, 512), (130, 512, 3000), (51200, 1500, 512), (37, 100, 68), (1, 4, 4)]


@pytest.mark.parametrize("rows,cols,K", SHAPES)
@pytest.mark.parametrize("wpc", [3, 4])
def test_every_unit_once_and_consistent_sharing(rows, cols, K, wpc):
    tiles = -(-rows // 128) * -(-cols // 128)
    nk = -(-K // 16)
    total = tiles * nk
    P = min(256 * wpc, max(1, total // 4), 8 * tiles)                    # xv_launch_gemm_nt, the evenly scheduled form
    total, begin, owner, shares = schedule(tiles, nk, P)
    covered = {}
    for w, tile, kt0, kt1, slot in shares:
        assert 0 <= kt0 < kt1 <= nk
        for kt in range(kt0, kt1):
            assert (tile, kt) not in covered
            covered[(tile, kt)] = w
    assert len(covered) == total
    # equal work: run lengths differ by at most one K-step
    lens = [begin[w + 1] - begin[w] for w in range(P)]
    assert max(lens) - min(lens) <= 1 and min(lens) >= 1
    # the owner formula is the inverse of the ranges
    for u in list(range(0, total, max(1, total // 997))) + [0, total - 1]:
        w = owner(u)
        assert begin[w] <= u < begin[w + 1]
    by_tile = {}
    for w, tile, kt0, kt1, slot in shares:
        by_tile.setdefault(tile, []).append((w, kt0, kt1, slot))
    slots = set()
    for tile, segs in by_tile.items():
        w_first, w_last = owner(tile * nk), owner(tile * nk + nk - 1)
        assert [s[0] for s in segs] == list(range(w_first, w_last + 1))   # the ticket count and the reducer's loop
        assert segs[0][1] == 0 and segs[-1][2] == nk and all(a[2] == b[1] for a, b in zip(segs, segs[1:]))   # K order
        if len(segs) > 1:
            for w, kt0, kt1, slot in segs:
                assert (w, slot) not in slots                            # a workgroup's two shared tiles never collide
                slots.add((w, slot))
                assert slot == (0 if tile == begin[w] // nk else 1)       # what the reducer recomputes for workgroup w
    # a tile is never split into more than a handful of shares (its last workgroup sums them alone)
    assert max(len(v) for v in by_tile.values()) <= 10
    # a workgroup has at most two shared tiles
    for w in range(P):
        assert sum(1 for (ww, tile, kt0, kt1, slot) in shares if ww == w and (kt0 != 0 or kt1 != nk)) <= 2


def run_order(tiles, nk, P, w):
    """The order in which workgroup w walks its run (xv_gemm_nt_sk_kernel, XV_SK_WRAP_FIRST): a run that ends one tile and begins the next
    takes the beginning of the next tile first, so that every workgroup starts near K = 0."""
    total = tiles * nk
    u0, u_end = w * total // P, (w + 1) * total // P
    u_mid = (u0 // nk + 1) * nk
    wrap_first = u0 % nk != 0 and u_mid < u_end and u_end - u_mid <= nk
    passes = [(u_mid, u_end), (u0, u_mid)] if wrap_first else [(u0, u_end)]
    segs = []
    for u, u_stop in passes:
        while u < u_stop:
            tile = u // nk
            kt0 = u - tile * nk
            kt1 = min(nk, kt0 + (u_stop - u))
            segs.append((tile, kt0, kt1))
            u += kt1 - kt0
    return segs, wrap_first


@pytest.mark.parametrize("rows,cols,K", SHAPES)
def test_wrap_first_order_covers_the_same_units(rows, cols, K):
    tiles = -(-rows // 128) * -(-cols // 128)
    nk = -(-K // 16)
    total = tiles * nk
    P = min(768, max(1, total // 4), 8 * tiles)
    _, begin, _, shares = schedule(tiles, nk, P)
    wrapped = 0
    for w in range(P):
        segs, wrap = run_order(tiles, nk, P, w)
        wrapped += wrap
        natural = sorted((tile, kt0, kt1) for ww, tile, kt0, kt1, slot in shares if ww == w)
        assert sorted(segs) == natural                                    # the same shares, another order
        if wrap:
            assert len(segs) == 2 and segs[0][1] == 0 and segs[0][0] == segs[1][0] + 1 and segs[1][2] == nk
            # in step with the others: the walk starts at K = 0 and its second leg starts no more than one run length further on
            assert segs[1][1] >= segs[0][2] - (begin[w + 1] - begin[w])
    if rows == 25088 and K == 2560:                                       # tdnn2's data gradient at S1: 784 tiles on 768 workgroups
        assert wrapped >= 700



# ---- "whole tiles + shares" in the one-workgroup-per-tile kernel (xv_nt_shares / xv_gemm_nt_kernel) -------------------------------------
def nt_shares(tiles, ksteps, stats, beside_wgrad, ws_bytes=1 << 40):
    """csrc/xv_gemm.hip xv_nt_shares, restated"""
    rem, whole = tiles % 256, tiles // 256
    if rem < 1 or rem > 128:
        return 0
    if not (whole >= 2 or (whole == 1 and stats)) or (beside_wgrad and whole == 3):
        return 0
    best, best_s = 0, 0
    sh = 2
    while sh <= 16 and ksteps // sh >= 6:
        t = whole * ksteps + -(-(rem * sh) // 256) * (ksteps // sh + 10) + sh
        if not best_s or t < best:
            best, best_s = t, sh
        sh += 1
    t_dp = -(-tiles // 256) * ksteps
    if not best_s or best + best // 32 >= t_dp or rem * best_s * 128 * 128 * 4 > ws_bytes:
        return 0
    return best_s


def test_share_plan_on_the_measured_shapes():
    """the decision table profiles/r04_nt_whole_plus_shares.txt was measured with"""
    t = lambda rows, cols: -(-rows // 128) * -(-cols // 128)
    # 64 x 300: tdnn2 / tdnn3 forward (584 / 572 tiles), the K = 512 layers, and their data gradients
    assert nt_shares(t(18688, 512), 160, True, False) == 3            # 72 remaining tiles: 216 shares, at most one per CU (4: 288 = two on some)
    assert nt_shares(t(18304, 512), 224, True, False) == 4            # 60 remaining tiles: 240 shares = 1 per CU
    assert nt_shares(t(18304, 512), 32, True, False) in (2, 3, 4, 5)
    assert nt_shares(t(18944, 512), 160, False, True) > 0
    # S1: every forward launch is balanced or too large; tdnn2's data gradient (784 tiles, beside the weight gradients) keeps the even schedule
    assert nt_shares(t(24576, 512), 160, True, False) == 0            # 768 tiles: no remainder
    assert nt_shares(t(25088, 512), 160, False, True) == 0
    assert nt_shares(t(25088, 512), 160, True, False) > 0             # the same tile count as a forward launch (64 x 400): shared
    assert nt_shares(t(23808, 1500), 32, True, False) == 0            # 2 232 tiles
    # one whole tile per CU: forward only; large remainders: never
    assert nt_shares(424, 224, False, True) == 0 and nt_shares(424, 224, True, False) == 0      # 168 remaining tiles
    assert nt_shares(300, 160, False, True) == 0 and nt_shares(300, 160, True, False) > 0
    assert nt_shares(664, 160, True, False) == 0                       # 152 remaining tiles
    assert nt_shares(100, 160, True, False) == 0 and nt_shares(1100, 160, True, False) > 0     # launches of several rounds: the shares fill the tail
    assert nt_shares(1544, 224, True, False) > 0 and nt_shares(1568, 160, False, True) > 0    # 128 x 400: tdnn3 forward (8 remaining tiles), tdnn2's data gradient
    # short K: a share is at least 6 K-steps
    assert nt_shares(532, 8, True, False) == 0
    assert nt_shares(584, 160, True, False, ws_bytes=1 << 20) == 0     # no room for the slabs


@pytest.mark.parametrize("tiles", [257, 300, 532, 572, 584, 640, 772, 784, 896, 1544, 1568, 4632])
@pytest.mark.parametrize("ksteps", [12, 32, 94, 160, 224])
def test_shares_cover_every_k_step_once(tiles, ksteps):
    sh = nt_shares(tiles, ksteps, True, False)
    if not sh:
        pytest.skip("schedule not used for this shape")
    n_whole = tiles // 256 * 256
    assert n_whole % 8 == 0                                            # both block ranges keep blockIdx % 8 = XCD
    cover = []
    for i in range(sh):                                                # xv_gemm_nt_kernel: share i = K-steps [i nk / S, (i + 1) nk / S)
        k0, k1 = i * ksteps // sh, (i + 1) * ksteps // sh
        assert k1 - k0 >= 6 or ksteps // sh >= 6
        cover += list(range(k0, k1))
    assert cover == list(range(ksteps))
    assert n_whole + (tiles - n_whole) * sh < 65536 * 16 and tiles <= 16384                  # XV_TN_MAX_TILES tickets


def test_library_reports_the_schedule_the_launcher_picks():
    """xv_debug_nt_schedule (include/xvector_hip.h): the launcher's own choice among one workgroup per tile (0), the even schedule (1),
    whole tiles + shares (2) and split-K (3), for the S1 and 64 x 300 layer shapes - what tools/pmc_traffic.py attributes layers to kernels
    with.  No GPU needed: it is host arithmetic."""
    from tf_kaldi_speaker_amd import _lib
    f = _lib.load().xv_debug_nt_schedule
    # S1 (128 x 200): every forward launch and every data gradient one workgroup per tile - tdnn2's data gradient (784 tiles: three whole tiles
    # per CU + 16) too since round 6: beside the weight-gradient stream a launch of >= 512 tiles never takes the even schedule; the same problem
    # with the chip to itself takes whole tiles + shares
    assert [f(24576, 512, 2560, 1, 0), f(23808, 512, 3584, 1, 0), f(23808, 512, 512, 1, 0), f(23808, 1500, 512, 1, 0)] == [0, 0, 0, 0]
    assert [f(25088, 512, 2560, 0, 1), f(24576, 512, 3584, 0, 1), f(23808, 512, 512, 0, 1), f(23808, 512, 1500, 0, 1)] == [0, 0, 0, 0]
    assert f(25088, 512, 2560, 0, 0) == 2
    # a small batch's data gradient (64 x 200: 392 tiles, 1.5 per CU) keeps the even schedule beside the weight gradient
    assert f(64 * 196, 512, 2560, 0, 1) == 1
    # 64 x 300: 584 / 572 tiles = two whole tiles per CU + 72 / 60 shared ones
    assert f(64 * 292, 512, 2560, 1, 0) == 2 and f(64 * 286, 512, 3584, 1, 0) == 2
    # one 300-frame utterance of an extraction run: 12 tiles, no statistics -> split-K
    assert f(286, 512, 3584, 0, 0) == 3
