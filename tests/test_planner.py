"""Product planner (smolscale_amd/csrc/smol_plan.c, through the C-ABI) against the
oracle's restatement of the reference planner and its sampling tables. CPU only."""
import ctypes
import random

import numpy as np

import cases as C
import helpers as H

FILTER_TO_REF = {H.FILTER_COPY: 0, H.FILTER_ONE: 1, H.FILTER_BOX: 9}


def ref_filter_id(f, halvings):
    return 2 + halvings if f == H.FILTER_BILINEAR else FILTER_TO_REF[f]


def check_against_oracle(t_in, w_in, h_in, t_out, w_out, h_out, srgb):
    job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    orc = H.OraclePlan()
    assert H.oracle().smol_oracle_plan(t_in, w_in, h_in, t_out, w_out, h_out, srgb, ctypes.byref(orc)) == 0
    key = (t_in, w_in, h_in, t_out, w_out, h_out, srgb)
    assert ref_filter_id(job.filter_h, job.halvings_h) == orc.filter_h, key
    assert ref_filter_id(job.filter_v, job.halvings_v) == orc.filter_v, key
    assert job.wide == (orc.storage_type == 3), key
    assert job.linear == orc.gamma_type, key
    assert job.premul16 == orc.premul16, key
    assert job.quirk24 == orc.quirk24, key
    if job.filter_h == H.FILTER_BOX:
        assert job.span_mul_x == orc.span_mul_x, key
    if job.filter_v == H.FILTER_BOX:
        assert job.span_mul_y == orc.span_mul_y, key
    return job


def test_planner_decisions_random():
    rng = random.Random(99)
    for _ in range(6000):
        def d():
            i = rng.choice([1, 2, 3, rng.randint(1, 70), rng.randint(1, 65535), 65535])
            o = rng.choice([1, 2, rng.randint(1, 70), rng.randint(1, 65535), max(1, i // 8), max(1, i // 255),
                            max(1, i // 8191), 65535])
            return i, o
        (w_in, w_out), (h_in, h_out) = d(), d()
        check_against_oracle(rng.randrange(10), w_in, h_in, rng.randrange(10), w_out, h_out, rng.randrange(2))


def test_planner_matches_golden_reference_tuples():
    rows = C.load_golden("planner.npz")["rows"]
    for r in rows:
        (i, o, t_in, t_out, srgb, fh, st_h, g_h, bil_w, hv_w, mul_x, fv, st_v, g_v, bil_h, hv_h, mul_y) = [int(v) for v in r]
        job = H.plan_job(t_in, i, 3, t_out, o, 3, srgb)
        assert ref_filter_id(job.filter_h, job.halvings_h) == fh, (i, o)
        assert job.wide == (st_h == 3) and job.linear == g_h
        if fh == 9:
            assert job.span_mul_x == mul_x
        if 2 <= fh <= 8:
            assert (o << job.halvings_h) == bil_w
        job = H.plan_job(t_in, 3, i, t_out, 3, o, srgb)
        assert ref_filter_id(job.filter_v, job.halvings_v) == fv, (i, o)
        assert job.wide == (st_v == 3) and job.linear == g_v
        if fv == 9:
            assert job.span_mul_y == mul_y


def test_baseline_config_filter_classes():
    """SURVEY 8a1 [measured]: none of the five BASELINE configs is a box filter."""
    j = H.plan_job(H.RGBA8_P, 640, 480, H.RGBA8_P, 320, 240)
    assert (j.filter_h, j.halvings_h, j.filter_v, j.halvings_v, j.wide) == (H.FILTER_BILINEAR, 0, H.FILTER_BILINEAR, 0, 0)
    j = H.plan_job(H.RGBA8_P, 4096, 4096, H.RGBA8_P, 1024, 1024)
    assert (j.filter_h, j.halvings_h, j.filter_v, j.halvings_v, j.wide) == (H.FILTER_BILINEAR, 1, H.FILTER_BILINEAR, 1, 0)
    j = H.plan_job(H.RGBA8_U, 1920, 1080, H.RGBA8_P, 7680, 4320)
    assert (j.filter_h, j.halvings_h, j.filter_v, j.halvings_v, j.wide) == (H.FILTER_BILINEAR, 0, H.FILTER_BILINEAR, 0, 0)
    j = H.plan_job(H.RGB8, 65535, 4096, H.RGBA8_P, 8192, 4096)
    assert (j.filter_h, j.halvings_h, j.filter_v, j.wide) == (H.FILTER_BILINEAR, 2, H.FILTER_COPY, 0)
    j = H.plan_job(H.RGBA8_P, 256, 256, H.RGBA8_P, 64, 64)
    assert (j.filter_h, j.halvings_h, j.filter_v, j.halvings_v) == (H.FILTER_BILINEAR, 1, H.FILTER_BILINEAR, 1)
    j = H.plan_job(H.RGBA8_P, 4096, 4096, H.RGBA8_P, 448, 448)
    assert (j.filter_h, j.filter_v, j.wide, j.span_mul_x) == (H.FILTER_BOX, H.FILTER_BOX, 0, 1836133)


def _closed_form_axis(job_step, job_frac0, filt, halv, dim_in, dim_out):
    """Expand the closed form into the reference's table layout (absolute offsets)."""
    lib = H.product()
    a, b, c, d = (ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32())
    out = []
    if filt == H.FILTER_BILINEAR:
        for j in range(dim_out << halv):
            lib.smol_hip_bilinear_sample(job_frac0, job_step, dim_in, j, ctypes.byref(a), ctypes.byref(b))
            out += [a.value, b.value]
    elif filt == H.FILTER_BOX:
        last = 0
        for k in range(dim_out):
            lib.smol_hip_box_span(job_step, dim_in, k, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d))
            out += [a.value, c.value]
            last = b.value
            if k:
                # left weight is 255 - previous F
                assert d.value == 255 - out[2 * (k - 1) + 1]
            else:
                assert d.value == 256
        out += [last, 0]
    return np.array(out, np.uint16)


def test_closed_form_sampling_equals_reference_tables():
    """The kernels never see the reference's uint16 tables; their closed form must
    expand to exactly those tables (vertical layout = absolute offsets)."""
    rng = random.Random(4)
    dims = [(640, 320), (4096, 1024), (1920, 7680), (1080, 4320), (65535, 8192), (4096, 448), (65535, 1),
            (65535, 65534), (2, 65535), (65534, 65535), (300, 7), (9, 1), (17, 2), (2, 2), (3, 2), (2, 3)]
    dims += [(rng.randint(2, 6000), rng.randint(1, 6000)) for _ in range(300)]
    dims += [(rng.randint(2, 65535), rng.randint(1, 65535)) for _ in range(40)]
    buf = np.zeros(140000, np.uint16)
    for (i, o) in dims:
        job = H.plan_job(0, 3, i, 0, 3, o, 0)
        if job.filter_v in (H.FILTER_COPY, H.FILTER_ONE):
            continue
        n = H.oracle().smol_oracle_precalc(3, i, 3, o, 1, buf.ctypes.data, buf.size)
        mine = _closed_form_axis(job.step_y, job.frac0_y, job.filter_v, job.halvings_v, i, o)
        assert n == mine.size, (i, o)
        assert np.array_equal(buf[:n], mine), (i, o)


def test_input_row_range_is_tight_and_sufficient():
    rng = random.Random(8)
    buf = np.zeros(140000, np.uint16)
    for _ in range(300):
        i, o = rng.randint(1, 3000), rng.randint(1, 3000)
        job = H.plan_job(0, 3, i, 0, 3, o, 0)
        first = rng.randint(0, o - 1)
        n = rng.randint(1, o - first)
        r0, r1 = H.sm.input_row_range(job, first, n)
        assert 0 <= r0 <= r1 <= i - 1
        if job.filter_v == H.FILTER_COPY:
            assert (r0, r1) == (first, first + n - 1)
        elif job.filter_v == H.FILTER_ONE:
            assert (r0, r1) == (0, 0)
        else:
            cnt = H.oracle().smol_oracle_precalc(3, i, 3, o, 1, buf.ctypes.data, buf.size)
            tab = buf[:cnt].reshape(-1, 2)
            if job.filter_v == H.FILTER_BILINEAR:
                h = job.halvings_v
                rows = tab[first << h:(first + n) << h, 0].astype(int)
                assert r0 == rows.min() and r1 == rows.max() + 1
            else:
                assert r0 == int(tab[first, 0]) and r1 == int(tab[first + n, 0])


def test_input_row_range_rejects_ranges_that_wrap_32_bits():
    """first_row + n_rows must not be evaluated modulo 2^32 (a wrapped sum once passed the check)."""
    import ctypes
    job = H.plan_job(0, 64, 64, 0, 32, 32, 0)
    a, b = ctypes.c_uint32(), ctypes.c_uint32()
    fn = H.product().smol_hip_input_row_range
    for first, n in ((0xFFFFFFFF, 2), (0xFFFFFFF0, 0x20), (32, 1), (31, 2), (0, 33), (0, 0)):
        assert fn(ctypes.byref(job), first, n, ctypes.byref(a), ctypes.byref(b)) == -10001, (first, n)
    assert fn(ctypes.byref(job), 31, 1, ctypes.byref(a), ctypes.byref(b)) == 0


def test_chunks_of_a_host_strip_take_the_road_the_strip_chose():
    """64bpp-class box x box picks the single pass (k_tile) or the two passes by the bytes a
    launch reads. The host road sizes the intermediate image from a probe of the whole strip and then launches the
    strip in chunks of ~8 MB: every chunk must take the probe's road (JobDev::box_force) -- on their own the chunks
    of a >= 150 MB source say "two passes" where the strip said k_tile, and find no intermediate image. Replayed
    on the dispatcher itself (tests/emu: strip_classify + classify per chunk), no GPU."""
    import ctypes
    emu = H.emu()
    emu.emu_strip_chunks_agree.restype = ctypes.c_int
    KC_TILE, KC_BOX2 = 3, 10
    seen = set()
    for (t_in, w_in, h_in, t_out, w_out, h_out) in (
            (0, 8000, 6000, 0, 800, 600),          # the advisor's three reproducers
            (0, 12000, 9000, 0, 1200, 900),
            (8, 16000, 12000, 8, 1000, 750),
            (0, 4096, 4096, 0, 448, 448),          # BOX row: one image is below the threshold -> two passes throughout
            (0, 16384, 16384, 0, 900, 900),        # ratio 18.2, 1 GiB
            (8, 4000, 3000, 8, 200, 150)):         # thumbnails: two passes whatever the size
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, 0)
        sp, dp = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
        # the road's own chunking: ~8 MB of traffic per chunk, at most 32 chunks
        total = h_in * sp + h_out * dp
        n_chunks = max(1, min(32, h_out, -(-total // (8 << 20))))
        rpc = -(-h_out // n_chunks)
        for first, n in ((0, h_out), (h_out // 3, h_out // 2)):
            k, flips = ctypes.c_int(), ctypes.c_int()
            rc = emu.emu_strip_chunks_agree(ctypes.byref(job), sp, dp, first, n, rpc, 1, ctypes.byref(k), ctypes.byref(flips))
            assert rc == 0 and flips.value == 0, (w_in, h_in, w_out, h_out, first, n, k.value, flips.value)
            assert k.value in (KC_TILE, KC_BOX2)
            seen.add(k.value)
            if (w_in, h_in) == (8000, 6000) and n == h_out:
                # ... and this is the hazard: without the verdict handed down, the chunks flip
                rc = emu.emu_strip_chunks_agree(ctypes.byref(job), sp, dp, first, n, rpc, 0, ctypes.byref(k), ctypes.byref(flips))
                assert k.value == KC_TILE and rc != 0 and flips.value > 0
    assert seen == {KC_TILE, KC_BOX2}


def test_chunks_of_a_host_strip_keep_the_splits_of_a_tall_box():
    """bilinear (COPY, ONE) H x tall box V in the 128bpp class keeps one 16-byte partial sum per output
    pixel and SPLIT of the row's span; twopass_setup takes more splits the fewer rows a call has, so the chunks of a strip
    would outgrow the intermediate image the strip's probe sized. The strip's count travels to its chunks
    (JobDev::hbox_force). Replayed on the dispatcher, no GPU: with the verdict every chunk fits, without it the
    advisor's reproducer does not."""
    import ctypes
    emu = H.emu()
    emu.emu_strip_chunks_agree.restype = ctypes.c_int
    KC_BOX2 = 10
    hazard = 0
    for (t_in, w_in, h_in, t_out, w_out, h_out, srgb) in (
            (5, 96, 32768, 5, 400, 512, 1),        # the advisor's reproducer: 1 split for the strip, 2 for 320-row chunks
            (4, 96, 32768, 4, 400, 512, 0),        # premul16
            (0, 96, 32768, 0, 400, 512, 1),
            (0, 300, 20000, 0, 200, 40, 0),        # compressed storage (ratio 500)
            (5, 64, 60000, 5, 100, 4, 1),
            (8, 640, 65535, 8, 641, 200, 1),
            (4, 2000, 40000, 5, 1500, 900, 1)):
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, srgb)
        sp, dp = w_in * H.bpp(t_in), w_out * H.bpp(t_out)
        for first, n in ((0, h_out), (h_out // 3, max(1, h_out // 2))):
            for rpc in sorted({1, 3, max(1, n // 7), max(1, n // 2), n, 320}):
                k, flips = ctypes.c_int(), ctypes.c_int()
                rc = emu.emu_strip_chunks_agree(ctypes.byref(job), sp, dp, first, n, rpc, 1, ctypes.byref(k), ctypes.byref(flips))
                assert k.value == KC_BOX2, (w_in, h_in, w_out, h_out, k.value)
                assert rc == 0 and flips.value == 0, (t_in, w_in, h_in, w_out, h_out, srgb, first, n, rpc, flips.value)
                rc = emu.emu_strip_chunks_agree(ctypes.byref(job), sp, dp, first, n, rpc, 0, ctypes.byref(k), ctypes.byref(flips))
                hazard += rc != 0
    assert hazard > 0


def test_mapped_road_chunks_fit_their_ring_slots():
    """The mapped road (smol_rt_lanes.hip: kernels on page-locked host rows) cuts a strip with a pageable side into chunks whose source
    rows / output rows pass through ONE ring slot each: the plan (smol_rt_chunks.h) replayed without a GPU over random jobs, row
    ranges, slot sizes and chunk sizes -- every chunk fits, the chunks tile the range, nothing outside the strip's source rows is
    read; page-locked on both sides is one launch; a row that cannot fit is reported (the caller takes the DMA road)."""
    emu = H.emu()
    u64 = ctypes.c_uint64
    emu.emu_mapped_chunks.restype = ctypes.c_int
    emu.emu_mapped_chunks.argtypes = [ctypes.POINTER(H.SmolHipJob)] + [ctypes.c_uint32] * 4 + [ctypes.c_int] * 2 + [u64, u64,
                                                                                                          ctypes.POINTER(u64), ctypes.POINTER(u64)]
    rng = random.Random(99)
    seen_plan = seen_none = 0
    for it in range(3000):
        t_in, t_out = rng.randrange(10), rng.randrange(10)
        w_in, h_in = rng.choice([1, 2, 17, 640, 4000, 65535]), rng.choice([1, 2, 33, 480, 3000, 20000])
        w_out, h_out = rng.choice([1, 3, 64, 333, 2048, 8192]), rng.choice([1, 2, 40, 299, 1024, 7000])
        job = H.plan_job(t_in, w_in, h_in, t_out, w_out, h_out, rng.randrange(2))
        first = rng.randrange(h_out)
        n = rng.randint(1, h_out - first)
        sp = w_in * H.bpp(t_in) + rng.choice([0, 5, 16])
        dp = w_out * H.bpp(t_out) + rng.choice([0, 3, 32])
        ring_in, ring_out = rng.randrange(2), rng.randrange(2)
        slot = rng.choice([64 << 10, 1 << 20, 8 << 20])
        want = rng.choice([64 << 10, 4 << 20, 8 << 20, 64 << 20])
        im, om = u64(), u64()
        rows = emu.emu_mapped_chunks(ctypes.byref(job), first, n, sp, dp, ring_in, ring_out, want, slot, ctypes.byref(im), ctypes.byref(om))
        assert rows >= 0, (rows, t_in, w_in, h_in, t_out, w_out, h_out, first, n, ring_in, ring_out, slot, want)
        if not ring_in and not ring_out:
            assert rows == n
        if rows:
            seen_plan += 1
            assert im.value <= slot and om.value <= slot
        else:
            seen_none += 1
            assert ring_in or ring_out          # only a row whose ring side exceeds the slot may be refused
    assert seen_plan > 2000 and seen_none > 10
