"""HIP cosine-similarity GEMM + fused top-1 vs the NumPy oracle (MatMul::calculate / getOutputs semantics)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mm(frt):
    m = frt.MatMul(0)
    yield m
    m.close()


@pytest.mark.parametrize("N,F", [(2, 1), (10000, 1), (10000, 4), (10007, 33), (12345, 128), (300, 200)])
def test_top1_and_full_matrix_match_oracle(frt, synth, mm, N, F):
    from oracle import match
    g = synth.make_gallery(N)
    r = np.random.Generator(np.random.PCG64(N + F))
    idx = r.integers(0, N, F)
    q = synth.make_queries(g, idx, noise=0.02)
    mm.init(g)
    got_i, got_s = mm.top1(q)
    want_i, want_s = match.top1(q, g)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_i, idx)
    assert np.abs(got_s - want_s).max() < 1e-5  # fp32, different summation order only
    full = mm.calculate(q)
    assert full.shape == (F, N)
    assert np.abs(full - match.similarity(q, g)).max() < 1e-5
    # the fused epilogue and the materialised matrix are the same arithmetic: bit-identical maxima
    assert np.array_equal(full.argmax(1), got_i) and np.array_equal(full.max(1), got_s)


def test_duplicate_rows_tie_break_first_index_wins(frt, synth, mm):
    g = synth.make_gallery(4096 + 37)
    for dst, src in ((4000, 5), (131, 130), (4100, 129), (2048, 2047)):
        g[dst] = g[src]
    q = np.concatenate([g[[5, 130, 129, 2047]], synth.make_queries(g, [4000, 131, 4100, 2048])])
    mm.init(g)
    i, s = mm.top1(q)
    assert i.tolist() == [5, 130, 129, 2047, 5, 130, 129, 2047]
    full = mm.calculate(q)
    assert np.array_equal(full[:, 4000], full[:, 5]) and np.array_equal(full[:, 2048], full[:, 2047])  # bit-identical duplicates


def test_reinit_is_idempotent_and_empty_gallery_is_an_error(frt, synth, mm):
    g1, g2 = synth.make_gallery(1000, seed=11), synth.make_gallery(2000, seed=12)
    mm.init(g1)
    mm.init(g2)  # /reload: the old device copy is released (the reference leaks it)
    q = synth.make_queries(g2, [1999])
    assert mm.top1(q)[0].tolist() == [1999]
    mm.init(g2[:0], 0, 512)
    with pytest.raises(frt.FrtError) as e:
        mm.top1(q)
    assert e.value.code == frt.FRT_ERR_EMPTY


def test_one_million_gallery_properties(frt, synth, mm):
    """BASELINE size (1M x 512 fp32 = 2.05 GB): planted answers, linearity, first-index rule; checked by properties."""
    N = 1_000_000
    g = synth.make_gallery(N)
    plant = np.array([0, 1, 127, 128, 499_999, 500_000, 999_871, 999_999])
    g[999_999] = g[127]  # duplicate at the far end: index 127 must win
    mm.init(g)
    q = synth.make_queries(g, plant, noise=0.01)
    i, s = mm.top1(q)
    want = plant.copy()
    want[-1] = 127
    assert np.array_equal(i, want) and s.min() > 0.97
    # scaling a query scales its similarity, the argmax is unchanged
    i2, s2 = mm.top1(2.0 * q)
    assert np.array_equal(i2, i) and np.allclose(s2, 2 * s, rtol=1e-6)
    # top-1 over the whole gallery == merge of top-1 over two halves (what the sharded multi-GPU path does)
    from oracle import match
    oi, osim = match.top1(q[:2], g)
    assert np.array_equal(oi, i[:2]) and np.abs(osim - s[:2]).max() < 1e-5


def test_screened_top1_is_bit_identical_to_the_exact_scan(frt, synth, mm):
    """N >= 32768 takes the fp16-screened path: it must return exactly what the full exact-fp32 scan returns, also when many
    rows sit inside the fp16 rounding band of the maximum, for non-unit norms, and for the first-index rule."""
    N = 40000 + 77
    r = np.random.Generator(np.random.PCG64(99))
    g = synth.make_gallery(N)
    g *= r.uniform(0.1, 30.0, (N, 1)).astype(np.float32)            # row norms from 0.1 to 30
    base = g[100].copy()
    for k, row in enumerate(range(20000, 20064)):                    # 64 near-duplicates of row 100: similarities within ~1e-4
        g[row] = base * np.float32(1 + 1e-5 * (k % 7)) + np.float32(1e-5) * r.standard_normal(512).astype(np.float32)
    g[39000] = g[123]                                                # exact duplicate far away: index 123 must win
    g[5] = 0                                                         # a zero row
    q = np.concatenate([synth.make_queries(g, [100, 123, 39000, 20010, 7], noise=0.0),
                        7.5 * synth.make_queries(g, [31000, 64, 40076], noise=0.05),  # non-unit query norms
                        r.standard_normal((24, 512)).astype(np.float32)])             # unrelated queries: tiny margins
    mm.init(g)
    i, s = mm.top1(q)
    full = mm.calculate(q)                                           # exact fp32 scan (full matrix)
    assert np.array_equal(i, full.argmax(1).astype(np.int32))
    assert np.array_equal(s, full.max(1))
    assert i[1] == 123 and i[2] == 123
    from oracle import match
    oi, osim = match.top1(q, g)
    agree = (oi == i)
    assert agree.mean() > 0.9 and np.abs(osim - s).max() < 2e-3 * np.abs(osim).max()  # NumPy sums in another order: near-ties may differ


def test_screening_switch_generation_and_scan_bytes(frt, synth, mm):
    """Round 5: frt_matcher_set_screening(m, 0) makes every top-1 call take the exact fp32 scan (bench.py's match_worst_case leg) - same
    answers bit for bit; frt_matcher_scan_bytes reports what a call reads in either mode; frt_matcher_generation moves with the gallery."""
    N = 50000
    g = synth.make_gallery(N)
    q = np.concatenate([synth.make_queries(g, [7, 49999, 1234], noise=0.01), np.random.Generator(np.random.PCG64(5)).standard_normal((29, 512)).astype(np.float32)])
    gen0 = int(frt.lib.frt_matcher_generation(mm._h))
    mm.init(g)
    gen1 = int(frt.lib.frt_matcher_generation(mm._h))
    assert gen1 != gen0
    assert mm.scanBytes() == N * 512            # int8 shadow: one byte per element
    i1, s1 = mm.top1(q)
    mm.setScreening(False)
    assert mm.scanBytes() == N * 512 * 4        # the stored fp32 rows
    i0, s0 = mm.top1(q)
    mm.setScreening(True)
    i2, s2 = mm.top1(q)
    assert np.array_equal(i0, i1) and np.array_equal(s0, s1) and np.array_equal(i2, i1) and np.array_equal(s2, s1)
    assert list(i1[:3]) == [7, 49999, 1234]
    mm.init(g[:100])
    assert int(frt.lib.frt_matcher_generation(mm._h)) != gen1 and mm.scanBytes() == 100 * 512 * 4


def test_sustained_mfma_probe_reports_a_plausible_rate(frt):
    """frt_probe_sustained_mfma (bench.py's roofline.sustained_peak): the three instruction mixes order as expected and stay below the nominal
    2.5 PFLOP/s."""
    a, b, c = (frt.probe_sustained_mfma(0, m, 0.05) for m in (0, 1, 2))
    assert 200 < c <= b * 1.05 and b <= a * 1.05 and a < 2600, (a, b, c)


@pytest.mark.parametrize("N", [5000, 70001])
def test_sharded_gallery_merge_equals_single_gallery(frt, synth, N):
    """Config 5 on one GPU: two matchers own disjoint row ranges (global indices via setRowOffset); the first-maximum merge of
    their winners must equal the single-gallery answer, duplicates across the shard boundary included."""
    g = synth.make_gallery(N)
    cut = N // 2 + 13
    g[cut + 5] = g[7]          # duplicate across shards: global index 7 wins
    g[cut - 1] = g[cut]        # duplicates straddling the boundary: cut-1 wins
    q = synth.make_queries(g, [7, cut + 5, cut, cut - 1, N - 1, 0], noise=0.0)
    whole, a, b = frt.MatMul(0), frt.MatMul(0), frt.MatMul(0)
    whole.init(g)
    a.init(g[:cut])
    b.init(g[cut:])
    b.setRowOffset(cut)
    wi, ws = whole.top1(q)
    ai, as_ = a.top1(q)
    bi, bs = b.top1(q)
    mi, ms = frt.merge_top1(ai, as_, bi, bs)
    assert np.array_equal(mi, wi) and np.array_equal(ms, ws)
    assert wi.tolist() == [7, 7, cut - 1, cut - 1, N - 1, 0]
    for m in (whole, a, b):
        m.close()


@pytest.mark.parametrize("D", [64, 128, 256, 512])
def test_screened_top1_at_every_supported_width(frt, D):
    """The coarse kernel is instantiated for D = 64 / 128 / 256 / 512 (the fp16 shadow gallery is stored in MFMA-fragment order, padded
    to whole 128-row tiles).  N not a multiple of 128, duplicate rows across tiles, random queries: screened == full matrix, bit for bit."""
    rng = np.random.default_rng(D)
    N = 50000 + 13
    g = rng.standard_normal((N, D)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g[45000] = g[77]
    g[N - 1] = g[N - 200]
    q = np.concatenate([g[[77, 45000, 12, N - 1, N - 200]], rng.standard_normal((60, D)).astype(np.float32)])
    m = frt.MatMul(0)
    m.init(g)
    i, s = m.top1(q)
    full = m.calculate(q)
    assert np.array_equal(i, full.argmax(1).astype(np.int32)) and np.array_equal(s, full.max(1))
    assert i[0] == 77 and i[1] == 77 and i[3] == N - 200 and i[4] == N - 200
    m.close()


def test_dim_not_multiple_of_64_stays_on_the_exact_scan(frt, synth):
    """the fp16 screening kernel walks K in steps of 64; D = 96 (and 32) with N >= 32768 must fall back to the exact
    scan instead of dropping the last 32 dimensions from the coarse scores."""
    from oracle import match
    rng = np.random.default_rng(5)
    for D in (96, 32):
        N, F = 40000, 37
        g = rng.standard_normal((N, D)).astype(np.float32)
        g /= np.linalg.norm(g, axis=1, keepdims=True)
        q = g[rng.integers(0, N, F)] + 0.05 * rng.standard_normal((F, D)).astype(np.float32)
        # make the LAST 32 dimensions decide: a decoy in another 128-row tile that beats the answer on the first 64 dims only (a coarse
        # pass blind to dims 64..95 would score it 1.0 against ~0.67 and drop the answer's tile from the exact re-rank)
        if D == 96:
            g[20000, :64] = 1.5 * g[7, :64]
            g[20000, 64:] = -g[7, 64:]
            q[0] = g[7]
        mm = frt.MatMul()
        mm.init(g)
        idx, sim = mm.top1(q)
        oi, osim = match.top1(q, g)
        assert np.array_equal(idx, oi), D
        assert np.abs(sim - osim).max() < 1e-5
        mm.close()


@pytest.mark.parametrize("N", [5000, 70000])
def test_fp16_stored_gallery(frt, synth, N):
    """BASELINE config 5 stores the gallery as fp16: similarities are then DEFINED on the fp16-rounded rows (fp32 accumulate).  Full
    matrix, top-1 (exact scan for N = 5000, screened for N = 70000) and duplicate-row ties against the NumPy oracle on those rows."""
    from oracle import match
    g = synth.make_gallery(N)
    g[N - 3] = g[11]          # duplicate rows: first index must win
    g16 = g.astype(np.float16).astype(np.float32)
    rng = np.random.default_rng(3)
    q = g[rng.integers(0, N, 45)] + 0.02 * rng.standard_normal((45, 512)).astype(np.float32)
    q[0] = g16[11]
    mm = frt.MatMul()
    mm.setStorage(True)
    mm.init(g)
    idx, sim = mm.top1(q)
    oi, osim = match.top1(q, g16)
    assert np.array_equal(idx, oi) and idx[0] == 11
    assert np.abs(sim - osim).max() < 1e-5
    full = mm.calculate(q[:5])
    assert np.abs(full - q[:5] @ g16.T).max() < 1e-5
    assert np.array_equal(full.argmax(1), idx[:5])
    # back to fp32 storage on the same object
    mm.setStorage(False)
    mm.init(g)
    idx32, sim32 = mm.top1(q)
    o32, s32 = match.top1(q, g)
    assert np.array_equal(idx32, o32) and np.abs(sim32 - s32).max() < 1e-5
    mm.close()


def test_streaming_gallery_load_equals_init(frt, synth):
    """initKnownEmbeds / addEmbedding x n / initMatMul as begin / append / commit: ragged appends (1 row, blobs, > one staging chunk),
    fewer rows than reserved, over-capacity error, reload on the same object."""
    from oracle import match
    N = 10000
    g = synth.make_gallery(N)
    q = g[[5, 4097, 9999]] + 0.01
    mm = frt.MatMul()
    mm.galleryBegin(N + 50, 512)
    mm.galleryAppend(g[0])                       # one row (addEmbedding)
    mm.galleryAppend(g[1:3].tobytes())           # a raw blob, as sqlite3_column_blob hands it over
    mm.galleryAppend(g[3:9000])                  # more than two staging chunks at once
    mm.galleryAppend(g[9000:])
    with pytest.raises(frt.FrtError) as e:
        mm.galleryAppend(np.zeros((51, 512), np.float32))
    assert e.value.code == frt.FRT_ERR_CAPACITY
    mm.galleryCommit()
    assert mm.m == N
    idx, sim = mm.top1(q)
    oi, osim = match.top1(q, g)
    assert np.array_equal(idx, oi) and np.abs(sim - osim).max() < 1e-5
    # /reload: the old gallery answers until commit
    mm.galleryBegin(100, 512)
    mm.galleryAppend(g[::-1][:100].copy())
    idx_mid, _ = mm.top1(q)
    assert np.array_equal(idx_mid, oi)
    mm.galleryCommit()
    idx2, _ = mm.top1(g[[9999, 9950]])
    assert list(idx2) == [0, 49]
    with pytest.raises(frt.FrtError):
        mm.galleryAppend(g[0])                   # no load in progress
    mm.close()


def test_screened_top1_pair_list_overflow_falls_back_to_the_exact_scan(frt, synth):
    """Round 3: the screened search re-ranks (query, tile) pairs with a scalar kernel; when more pairs qualify than its list holds - a
    gallery of identical rows puts EVERY tile within the rounding band of the maximum, a NaN query has no bound at all - the call is
    answered by the unscreened exact scan instead.  Same answers as the materialised matrix, first index on ties."""
    N = 50000
    g = np.tile(synth.make_gallery(1), (N, 1))                      # all rows equal: every row attains the maximum, index 0 must win
    q = np.concatenate([g[:1], synth.make_gallery(47, seed=5)])
    m = frt.MatMul(0)
    m.init(g)
    i, s = m.top1(q)                                                # 48 queries x 391 tiles = 18 768 candidate pairs > the list's capacity
    full = m.calculate(q)
    assert np.array_equal(i, np.zeros(48, np.int32)) and np.array_equal(s, full.max(1)) and np.array_equal(full.argmax(1), i)
    # the next call on the same object is an ordinary one again (the overflow flag is cleared per call)
    g2 = synth.make_gallery(N)
    g2[40000] = g2[9]
    m.init(g2)
    q2 = synth.make_queries(g2, [9, 40000, 777, 49999])
    i2, s2 = m.top1(q2)
    assert i2.tolist() == [9, 9, 777, 49999]
    # a NaN / inf query: no rounding bound -> every tile -> fallback; the finite queries of the same call still get their exact answers
    q3 = q2.copy()
    q3[1, 5] = np.nan
    q3[2, 7] = np.inf
    i3, s3 = m.top1(q3)
    full3 = m.calculate(q3)
    assert i3[0] == 9 and i3[3] == 49999 and np.array_equal(s3[[0, 3]], full3.max(1)[[0, 3]])
    assert i3[1] == -1                                              # every similarity of a NaN query is NaN: nothing is ever "greater"
    m.close()


def test_screened_top1_one_overflowing_pair_sub_list_falls_back_to_the_exact_scan(frt, synth):
    """Round 6: the pair list is 64 sub-lists (tile segment x query & 3), each with its own counter on its own cache line.  ONE sub-list
    overflowing - six queries of the same class that each match a planted row in all 32 tiles of segment 0: 192 pairs for 128 slots, with the
    list as a whole almost empty - must send the call through the exact scan like a full list does: same answers as the materialised matrix,
    first index among the identical planted rows."""
    N = 65536                                                       # 512 tiles: 32 per segment
    g = synth.make_gallery(N)
    d = synth.make_gallery(1, seed=11)[0]
    planted = np.arange(32) * 128 + 5                               # one row in every tile of segment 0
    g[planted] = d
    q = synth.make_gallery(24, seed=7)
    same_class = [0, 4, 8, 12, 16, 20]                              # query & 3 == 0
    q[same_class] = d
    m = frt.MatMul(0)
    m.init(g)
    i, s = m.top1(q)
    full = m.calculate(q)
    assert np.array_equal(i, full.argmax(1).astype(np.int32)) and np.array_equal(s, full.max(1))
    assert all(int(i[k]) == 5 for k in same_class)                  # the first of the identical rows
    # and the same object answers an ordinary call afterwards (flag and counters are cleared per call)
    q2 = synth.make_queries(g, [9, 40000, 777, 65535])
    i2, _ = m.top1(q2)
    assert i2.tolist() == [9, 40000, 777, 65535]
    m.close()


def test_calculate_top1_is_the_matrix_and_its_row_maxima_in_one_call(frt, synth):
    """frt_matcher_calculate_top1 (what the drop-in ArcFaceIR50::featureMatching + getOutputs use): the materialised [F, N] matrix equals
    calculate(), the (idx, sim) pairs equal top1() AND std::max_element over the rows, bit for bit; without the matrix the pairs are the same."""
    for N, fp16 in ((5000, False), (60000, False), (60000, True)):
        g = synth.make_gallery(N)
        g[N - 7] = g[3]
        q = np.concatenate([g[[3]], synth.make_queries(g, [N - 1, 1234, 77], noise=0.05)])
        m = frt.MatMul(0)
        m.setStorage(fp16)
        m.init(g)
        full, i, s = m.calculate_top1(q)
        assert np.array_equal(full, m.calculate(q))
        assert np.array_equal(i, full.argmax(1).astype(np.int32)) and np.array_equal(s, full.max(1)) and i[0] == 3
        _, i2, s2 = m.calculate_top1(q, materialize=False)
        i3, s3 = m.top1(q)
        assert np.array_equal(i2, i) and np.array_equal(s2, s) and np.array_equal(i3, i) and np.array_equal(s3, s)
        m.close()


@pytest.mark.gpu
def test_int8_shadow_screening_is_exact_whatever_the_rows_look_like(frt, synth):
    """Round 4: fp32-stored 512-column galleries are screened through an INT8 shadow (per-row scale, measured error norm in the bound,
    kernels_match.hip).  Rows that stress the quantiser - one huge element (everything else rounds to 0), tiny rows, all-zero rows,
    un-normalised rows 3x longer than the rest, near-duplicates whose difference is far below one int8 step, exact duplicates across
    tiles - and queries that match nothing (widest candidate band) must still give the full-matrix scan's answer bit for bit; a NaN /
    inf row voids the bound and the call must take the exact scan and still agree."""
    rng = np.random.default_rng(8)
    N = 70000 + 77
    g = synth.make_gallery(N).copy()
    g[100] = 0
    g[100, 3] = 1.0                              # one-hot: scale = 1/127, every other element quantises to 0
    g[200] *= 1e-6                               # tiny row
    g[300] = 0                                   # all-zero row (scale 0)
    g[400:410] *= 3.0                            # un-normalised rows dominate max||g|| and the error norm
    g[5000] = g[64000]                           # exact duplicates 59 000 rows apart: first index wins
    g[6000] = g[64001] * (1 + 1e-6)              # near-duplicate: the int8 rows are identical, the exact re-rank separates them
    q = np.concatenate([g[[100, 200, 64000, 64001, 6000, 405]], rng.standard_normal((58, 512)).astype(np.float32)])
    q[:6] /= np.maximum(np.linalg.norm(q[:6], axis=1, keepdims=True), 1e-30)
    q[6:] /= np.linalg.norm(q[6:], axis=1, keepdims=True)
    m = frt.MatMul(0)
    m.init(g)
    i, s = m.top1(q)
    full = m.calculate(q)
    assert np.array_equal(i, full.argmax(1).astype(np.int32)) and np.array_equal(s, full.max(1))
    assert i[2] == 5000 and i[1] == full[1].argmax()
    ti, ts = m.topk(q, 4)
    order = np.argsort(-full, axis=1, kind="stable")[:, :4]
    assert np.array_equal(ti, order.astype(np.int32)) and np.array_equal(ts, np.take_along_axis(full, order, 1))
    # a non-finite row: no bound -> every call goes through the exact scan (NaN never wins, std::max_element's rule)
    g2 = g.copy()
    g2[12345, 7] = np.nan
    g2[23456, 9] = np.inf
    m.init(g2)
    i2, s2 = m.top1(q[:8])
    full2 = m.calculate(q[:8])
    want = np.array([np.nanargmax(np.where(np.isnan(r), -np.inf, r)) for r in full2], np.int32)
    assert np.array_equal(i2, want)
    m.close()
