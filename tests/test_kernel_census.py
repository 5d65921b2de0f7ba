"""CPU guard on the shape of the hot loop: the gfx950 listing of the production ZBC kernel (hipcc cross-compiles here)
must keep the instruction counts DESIGN.md section 5 and bench.py's VALU_MODEL quote -- a compiler or source change
that adds wide multiplies, transcendentals, scratch or registers shows up here without a GPU."""
import os
import re
import subprocess
import sys
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd")


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "eng.s")
    import hwmc
    flags = [f for f in hwmc.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]        # the PRODUCTION flags (build.py), device listing only
    assert "-fno-slp-vectorize" in flags and "--offload-arch=gfx950" in flags
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", "-o", out,
                           os.path.join(PKG, "csrc", "hw_engine.hip")], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel(listing, pattern):
    m = re.search(r"^(_ZN3hwk[^\n:]*%s[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % re.escape(pattern), listing, re.S | re.M)
    assert m, pattern
    tail = listing[m.end():m.end() + 6000]
    meta = {k: int(re.search(r"; %s: (\d+)" % k, tail).group(1)) for k in ("NumVgprs", "ScratchSize", "Occupancy")}
    return m.group(2).split("\n"), meta


def _hot_loops(lines):
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if mm:
            labels[mm.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
            spans.append((labels[mm.group(1)], i))
    inner = [a for a in spans if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in spans)]
    out = []
    for a, b in inner:
        ins = [l.strip().split()[0] for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
        c = Counter(ins)
        if c["v_mad_u64_u32"] >= 10:
            out.append(c)
    return out


def test_zbc_f32_hot_loop_keeps_its_instruction_budget(listing):
    lines, meta = _kernel(listing, "zbc_superposed_kernelIfLi1ELi1E")
    assert meta["ScratchSize"] == 0 and meta["NumVgprs"] <= 64 and meta["Occupancy"] == 8
    loops = _hot_loops(lines)
    trans = ("v_log_f32_e32", "v_sqrt_f32_e64", "v_sqrt_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32")
    best = min(loops, key=lambda c: (c["v_mad_u64_u32"], sum(v for k, v in c.items() if k.startswith("v_"))))
    valu = sum(v for k, v in best.items() if k.startswith("v_"))
    assert best["v_mad_u64_u32"] == 16                      # scalar round-2 multiply (UHI copy of the loop)
    assert best["v_bitop3_b32"] == 14 and best["v_xor_b32_e32"] <= 4
    assert sum(best[t] for t in trans) == 8 and best["v_cvt_f32_u32_e32"] == 4
    assert valu <= 60, (valu, dict(best))                   # 60 today: 32 single-rate + 20 half-rate + 8 transcendentals
    # every copy of the blocked loop (with and without the uniform high word) stays within one multiply of it
    assert all(c["v_mad_u64_u32"] <= 17 for c in loops)


def test_zbc_f64_hot_loop_keeps_its_instruction_budget(listing):
    """The double stepper in the (y, W) block form: per Philox block 3 v_cvt_f64_f32 + 2 v_fma(c)_f64 + 1 v_add_f64 (the
    per-step form it replaced had 4 + 4 + 4), the same 16 multiplies, 8 transcendentals and 4 conversions as the float
    loop, and at most 30 single-rate instructions: bench.py's VALU_MODEL["superposed_f64"] = 30 / 26 / 8."""
    lines, meta = _kernel(listing, "zbc_superposed_kernelIdLi1ELi1E")
    # 62 VGPRs since the epilogue's exp2 goes through the float unit (hw_exp2(double); ocml's exp2 cost registers and
    # 106 SGPRs: 7 waves per SIMD then): 8 waves per SIMD.  0.202 ms at 2^20 pairs = the float kernel's 0.187 ms x 228 / 208
    # model cycles less what the shorter epilogue saves.
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] == 8 and meta["NumVgprs"] <= 64
    loops = _hot_loops(lines)
    best = min(loops, key=lambda c: (c["v_mad_u64_u32"], sum(v for k, v in c.items() if k.startswith("v_"))))
    trans = ("v_log_f32_e32", "v_sqrt_f32_e64", "v_sqrt_f32_e32", "v_sin_f32_e32", "v_cos_f32_e32")
    f64 = {k: v for k, v in best.items() if "f64" in k}
    assert best["v_mad_u64_u32"] == 16 and sum(best[t] for t in trans) == 8 and best["v_cvt_f32_u32_e32"] == 4
    assert best["v_cvt_f64_f32_e32"] == 3, f64
    assert best["v_fmac_f64_e32"] + best["v_fma_f64"] == 2 and best["v_add_f64"] == 1, f64
    half = 16 + 4 + 3 + 2 + 1
    valu = sum(v for k, v in best.items() if k.startswith("v_"))
    assert valu - half - 8 <= 30, (valu, dict(best))         # single-rate rest: 30 today


def test_curve_kernel_reduces_sixteen_maturities_together(listing):
    """The blocked float curve kernel: no LDS traffic and no wave-wide reduction per maturity -- the transposed tree
    (MaturityTree) spends 2 v_cndmask + 1 v_add_f32_dpp per merge, four merge levels in the loop body, the pending
    vectors updated in place (no v_mov storm around the asm: at most 6 v_mov in the whole loop body of the P = 1 kernel),
    and exactly one pair of v_exp_f32 per half-block event."""
    lines, meta = _kernel(listing, "zcb_curve_blocked_kernelILi1EfE")
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] == 8 and meta["NumVgprs"] <= 56
    body = [l.strip().split()[0] for l in lines if l.strip() and not l.strip().startswith((";", "."))]
    c = Counter(body)
    assert c["v_add_f32_dpp"] == 12                           # 4 levels x (2 copies of the loop: uniform high word or not, + the tail that completes the last group)
    assert c["v_exp_f32_e32"] + c["v_exp_f32_e64"] == 8       # 2 events x 2 exponentials x 2 copies
    loops = _hot_loops_with_branches(lines)
    assert loops and all(l["v_mov_b32_e32"] + l["v_mov_b32"] <= 6 for l in loops), loops


def _hot_loops_with_branches(lines):
    """Loop bodies (back edge to a label >= 150 lines earlier) that contain the Philox multiplies."""
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if mm:
            labels[mm.group(1)] = i
    out = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and 150 < i - labels[mm.group(1)] < 400:
            ins = [x.strip().split()[0] for x in lines[labels[mm.group(1)]:i + 1] if x.strip() and not x.strip().startswith((";", "."))]
            c = Counter(ins)
            if 14 <= c["v_mad_u64_u32"] <= 18:
                out.append(c)
    return out


def test_reference_stepper_loop_walks_a_whole_philox_block_per_trip(listing):
    """zbc_reference_kernel (the reference's per-step arithmetic verbatim on both legs): the four drifts of a block come
    with ONE wave-uniform s_load_dwordx4 (the per-step version issued a scalar load + wait + bounds branch + v_mov per
    step and ran 11 % slower), and the loop body holds 58 single-rate + 20 half-rate + 8 quarter-rate instructions =
    bench.py's VALU_MODEL["reference"]."""
    lines, meta = _kernel(listing, "zbc_reference_kernelILi0E")
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] == 8
    loops = _hot_loops(lines)
    best = min(loops, key=lambda c: (c["v_mad_u64_u32"], sum(v for k, v in c.items() if k.startswith("v_"))))
    assert best["s_load_dwordx4"] == 1 and best["s_load_dword"] == 0
    assert best["v_mad_u64_u32"] == 16 and best["v_fmac_f32_e32"] + best["v_fma_f32"] == 24 and best["v_add_f32_e32"] == 8
    assert best["v_mov_b32_e32"] + best["v_mov_b64_e32"] == 0
    valu = sum(v for k, v in best.items() if k.startswith("v_"))
    assert valu <= 86, (valu, dict(best))


def test_two_pairs_in_flight_loop_is_interleaved_and_unpacked(listing):
    """zbc_superposed_kernel<float, 1, 2> (two pairs per thread, used from two chip-fills of pairs up): exactly twice the
    one-pair loop's instructions -- 32 multiplies, 16 transcendentals -- and NO v_pk_* / v_mov in it (-fno-slp-vectorize:
    the SLP vectoriser otherwise packs the two pairs' steps and pays 12 v_mov per block for it)."""
    lines, meta = _kernel(listing, "zbc_superposed_kernelIfLi1ELi2E")
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] >= 6
    loops = _hot_loops_x2(lines)
    best = min(loops, key=lambda c: (c["v_mad_u64_u32"], sum(v for k, v in c.items() if k.startswith("v_"))))
    assert best["v_mad_u64_u32"] == 32 and best["v_bitop3_b32"] == 28
    assert sum(v for k, v in best.items() if k.startswith("v_pk_")) == 0 and best["v_mov_b32_e32"] == 0
    valu = sum(v for k, v in best.items() if k.startswith("v_"))
    assert valu <= 2 * 60, (valu, dict(best))


def _hot_loops_x2(lines):
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if mm:
            labels[mm.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
            spans.append((labels[mm.group(1)], i))
    inner = [a for a in spans if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in spans)]
    out = []
    for a, b in inner:
        ins = [l.strip().split()[0] for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
        c = Counter(ins)
        if 30 <= c["v_mad_u64_u32"] <= 34:
            out.append(c)
    return out


def test_other_kernels_have_no_scratch_and_full_occupancy(listing):
    for pat in ("vega_pathwise_kernelIfLi1E", "zcb_curve_blocked_kernelILi1EfE", "zcb_curve_blocked_kernelILi2EfE",
                "zbc_superposed_kernelIfLi2ELi1E", "zbc_reference_kernelILi0E"):
        _, meta = _kernel(listing, pat)
        assert meta["ScratchSize"] == 0 and meta["Occupancy"] == 8, (pat, meta)
    _, meta = _kernel(listing, "zbc_superposed_kernelIfLi3ELi1E")      # 15 double accumulators: 7 waves per SIMD
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] >= 7, meta
    # four pairs per thread carry 4 x (4 Philox words + y + W + live): 74 registers, 6 waves per SIMD -- measured
    # faster than two pairs at 8 waves all the same (tools/labs/curve_p_probe.py); it must not spill or drop further
    _, meta = _kernel(listing, "zcb_curve_blocked_kernelILi4EfE")
    assert meta["ScratchSize"] == 0 and meta["Occupancy"] >= 6 and meta["NumVgprs"] <= 80, meta
    # the double engine: state in double costs registers (y, W take two each) but no occupancy class
    for pat, occ in (("zcb_curve_blocked_kernelILi1EdE", 8), ("zcb_curve_blocked_kernelILi2EdE", 8), ("zcb_curve_blocked_kernelILi4EdE", 6),
                     ("vega_pathwise_kernelIdLi1E", 8), ("zbc_superposed_kernelIdLi3ELi1E", 6), ("zcb_curve_kernelIdE", 8)):
        _, meta = _kernel(listing, pat)
        assert meta["ScratchSize"] == 0 and meta["Occupancy"] >= occ, (pat, meta)


def test_vega_kernel_runs_the_same_time_loop_as_the_zbc_kernel(listing):
    """bench.py quotes the measured issue bound of the ZBC loop for the pathwise-Vega kernel too (q3_vega_gpu.valu): both
    kernels call the same ou_walk, and their hot loops must hold the same instructions, opcode by opcode."""
    def best_loop(pattern):
        lines, _ = _kernel(listing, pattern)
        loops = _hot_loops(lines)
        c = min(loops, key=lambda c: (c["v_mad_u64_u32"], sum(v for k, v in c.items() if k.startswith("v_"))))
        return {k: v for k, v in c.items() if k.startswith("v_")}
    assert best_loop("vega_pathwise_kernelIfLi1E") == best_loop("zbc_superposed_kernelIfLi1ELi1E")


def test_the_issue_bound_probe_times_the_production_loop():
    """bench.py's valu.bound_cycles_per_block comes from tools/mix_bound.hip, whose loop bodies (tools/mix_bound_body.inc)
    are GENERATED from the production kernel's gfx950 listing with the data dependencies removed (tools/gen_mix_bound.py).
    The committed .inc must be what the current sources generate -- a change of the hot loop that is not followed by a
    new probe run shows up here, without a GPU -- and the census the bound is quoted for is the loop's: 32 single-rate,
    20 half-rate (16 v_mad_u64_u32 + 4 v_cvt_f32_u32), 8 quarter-rate instructions per Philox block, twice that for the
    two-pairs-per-thread loop; the REAL variants are the production loops verbatim."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mix_bound
    text = gen_mix_bound.generate()
    committed = open(os.path.join(ROOT, "tools", "mix_bound_body.inc")).read()
    assert text == committed, "tools/mix_bound_body.inc is stale: python tools/gen_mix_bound.py, then re-run tools/bin/mix_bound on the GPU"
    for name, val in (("PROD1_FULL", 32), ("PROD1_HALF", 20), ("PROD1_QUARTER", 8), ("PROD2_FULL", 64), ("PROD2_HALF", 40), ("PROD2_QUARTER", 16)):
        assert re.search(r"#define MIX_CENSUS_%s %d\n" % (name, val), committed), name
    # a dependency-free twin never reads what the loop writes: sources v0-v7 (+ SGPRs, literals), destinations v8-v39
    body = committed.split("#define MIX_BODY_PROD1 ")[1].split("#define MIX_CLOBBERS_PROD1")[0]
    n_valu = 0
    for line in re.findall(r'"(v_[^"]*?)\\n', body):
        ops = [o.strip() for o in line[len(line.split()[0]):].split(",")]
        dest = [int(x) for x in re.findall(r"\d+", ops[0].split(":")[0])]
        assert 8 <= dest[0] <= 39, line
        first_src = 2 if line.startswith("v_mad_u64_u32") else 1
        for o in ops[first_src:]:
            for m in re.finditer(r"\bv\[?(\d+)", o):
                assert int(m.group(1)) <= 7, line
        n_valu += 1
    assert n_valu == 60
    # the committed measurement was taken for this census
    import json
    prof = json.load(open(os.path.join(ROOT, "profiles", "r04_mix_bound.json")))
    assert prof["census_per_block"] == {"full": 32, "half": 20, "quarter": 8}
    assert 144.0 <= prof["exclusive_classes_measured"] < prof["bound_cycles_per_block"] <= prof["nominal_serial_model"] == 208
    assert abs(prof["real_loop_cycles_per_block"]["one_pair_per_thread"] - prof["bound_cycles_per_block"]) < 1.0


def test_the_other_two_production_loops_have_twins_of_their_own():
    """the double ZBC loop and the curve loop are priced against THEIR OWN measured issue bound, not the
    float ZBC loop's.  Pinned here, without a GPU, from the generated twin bodies (the staleness of the whole file is the test
    above): the double loop's census (the float loop's 16 wide multiplies, 8 transcendentals and bitwise rounds + 3 v_cvt_f64_f32,
    2 v_fma_f64 and 1 v_add_f64 per block), and the curve twins' structure -- one trip = one period of sixteen maturities, with
    exactly the trips the event logic of zcb_curve_blocked_kernel takes in it (1000-step grid: 24 event-free trips, 8 with a
    maturity after the first half block, 8 after the second; 4000-step grid: 144 + 16), every butterfly depth 0..4 as often as
    trailing-ones(m & 15) says, ONE flush; two v_exp_f32 per event and pair; no LDS access, compare or branch of the production
    loop left in a twin (they can only lower a bound)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mix_bound as g
    committed = open(os.path.join(ROOT, "tools", "mix_bound_body.inc")).read()

    def body(name):
        txt = committed.split("#define MIX_BODY_%s " % name)[1].split("#define MIX_CLOBBERS_%s" % name)[0]
        return re.findall(r'"([^"]*?)\\n', txt)
    f64 = body("F64_1")
    ops = [l.split()[0] for l in f64]
    count = lambda pre: sum(o.startswith(pre) for o in ops)
    assert count("v_mad_u64_u32") == 16 and count("v_cvt_f32_u32") == 4 and count("v_cvt_f64_f32") == 3
    assert count("v_fma_f64") + count("v_fmac_f64") == 2 and count("v_add_f64") == 1
    assert sum(count(t) for t in ("v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")) == 8 and count("v_exp") == 0
    assert g.census_of(f64) == {"full": 30, "half": 26, "quarter": 8}
    assert [l for l in body("REALF64_1") if l.startswith("v_")] != [l for l in f64 if l.startswith("v_")]      # (REAL: real registers)
    assert [l.split()[0] for l in body("REALF64_1") if l.startswith("v_")] == [o for o in ops if o.startswith("v_")]
    # the event logic, restated by gen_mix_bound.curve_period and checked here against first principles
    for n_steps, n_events_1, n_events_2, n_trips in ((1000, 8, 8, 40), (4000, 0, 16, 160)):
        units, trips = g.curve_period(n_steps)
        kinds = [k for u in units for k, _ in u]
        assert trips == n_trips == len(kinds) == 16 * (n_steps // 100) // 4
        assert sum(k[0] == 1 for k in kinds) == n_events_1 and sum(k[0] == 2 for k in kinds) == n_events_2
        levels = sorted(k[1] for k in kinds if k[0])
        assert levels == sorted(min(4, (~m & (m + 1)).bit_length() - 1) for m in range(1, 17))   # trailing ones of m & 15, capped at 4
        assert sum(k[2] for k in kinds) == 1
    for name, P, n_trips, n_events in (("CURVE1", 1, 40, 16), ("CURVE2", 2, 160, 16), ("CURVE2_F64", 2, 160, 16)):
        lines = body(name)
        assert re.search(r"#define MIX_BLOCKS_%s %d\n" % (name, n_trips * P), committed)
        assert not [l for l in lines if l.startswith(("ds_", "s_waitcnt", "s_bitcmp"))]
        branches = [l for l in lines if l.startswith(("s_cbranch", "s_cmp", "s_branch"))]
        assert all(re.match(r"^(s_cmp_lg_u32 s7[23], 0|s_cbranch_scc1 \d+b)$", l) for l in branches), branches[:3]   # the twin's own counted loops only
        # executed instruction counts over one trip of the twin: unroll its counted loops
        def executed(lines):
            stack, out = [], {}
            counts, i = {}, 0
            mult = 1
            frames = []
            for l in lines:
                m = re.match(r"^s_mov_b32 (s7[23]), (\d+)$", l)
                if m:
                    frames.append(int(m.group(2)))
                    mult *= frames[-1]
                    continue
                if re.match(r"^s_cbranch_scc1 \d+b$", l):
                    mult //= frames.pop()
                    continue
                op = l.split()[0]
                counts[op] = counts.get(op, 0) + mult
            return counts
        ex = executed(lines)
        assert ex["v_mad_u64_u32"] == 16 * P * n_trips
        assert ex["v_exp_f32_e32"] + ex["v_exp_f32_e64"] == 2 * P * n_events
        assert ex["v_add_f32_dpp"] == sum(min(4, (~m & (m + 1)).bit_length() - 1) for m in range(1, 17))   # 15 butterfly merges per 16 maturities
        assert ex.get("v_cvt_f64_f32_e32", 0) >= 1                                                     # the flush's conversion (+ the double state's)
    # the committed measurement: every family has a bound, the straight-line loops reproduce their own rate in the harness,
    # and the curve loops cost more than the ZBC loop they contain
    import json
    loops = json.load(open(os.path.join(ROOT, "profiles", "r04_mix_bound.json")))["loops"]
    assert set(loops) >= {"zbc_f32", "zbc_f64", "curve_f32_p1", "curve_f32_p2", "curve_f64_p2"}
    assert abs(loops["zbc_f64"]["real_loop_cycles_per_block"] - loops["zbc_f64"]["bound_cycles_per_block"]) < 1.0
    assert loops["zbc_f32"]["bound_cycles_per_block"] < loops["curve_f32_p2"]["bound_cycles_per_block"] < loops["curve_f32_p1"]["bound_cycles_per_block"]
    assert loops["zbc_f32"]["bound_cycles_per_block"] < loops["zbc_f64"]["bound_cycles_per_block"] < loops["curve_f64_p2"]["bound_cycles_per_block"]
