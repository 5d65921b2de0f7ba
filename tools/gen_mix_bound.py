#!/usr/bin/env python3
"""Generate tools/mix_bound_body.inc: the hot loop of the PRODUCTION ZBC kernel, instruction for instruction, with its data
dependencies removed -- the input of tools/mix_bound.hip, which times it on the GPU to obtain the issue bound of the
loop's instruction mix (bench.py `valu`).

    python tools/gen_mix_bound.py            # compiles the engine's device listing with the production flags, rewrites the .inc
    python tools/gen_mix_bound.py --check    # exit 1 if the committed .inc is not what the current sources generate
    python tools/gen_mix_bound.py --search 96   # tools/mix_search_body.inc (NOT committed: 7 500 generated lines, fixed seed):
                                                # hipcc ... -DMIX_INC='"mix_search_body.inc"' tools/mix_bound.hip -o tools/bin/mix_search

How a dependency is removed: every VALU instruction keeps its opcode, encoding (e32/e64), modifiers, literal constants and
the CLASS of every operand (VGPR / SGPR / literal), in the production order, between the production's own s_setprio /
s_nop; but every VGPR source is redirected to one of eight input registers v0-v7 that nothing in the loop writes (same
register bank, index mod 4, as the production operand), and every VGPR destination to the next register of a rotating pool
v8-v39 that nothing in the loop reads (v_fmac's accumulator, the only read-modify-write, meets its previous value 32
destinations later).  No instruction therefore ever waits for another one's result: what remains is what the SIMD
needs to ISSUE this mix.  Scalar ALU instructions of the loop are kept with their registers renamed into a scratch range;
the loop counter and branch are the harness's own.

Variants (each a C string macro MIX_BODY_<NAME>, with MIX_BLOCKS_<NAME> Philox blocks per trip and MIX_CLOBBERS_<NAME>):
  PROD1 / PROD2          the loop of zbc_superposed_kernel<float,1,1> / <float,1,2> (one / two pairs per thread), the copy
                         taken when the path id's high word is wave-uniform (16 wide multiplies per block) -- the
                         production order and priorities, dependencies removed
  PROD1_NOPRIO / PROD2_NOPRIO   the same without s_setprio / s_nop
  GROUPED / GROUPED_PRIO   PROD1's instructions sorted by class (multiplies, conversions, transcendentals, single-rate)
  EXCL                   only the exclusive classes of PROD1 (16 v_mad_u64_u32, 4 v_cvt, 8 transcendentals), production order
  SIMPLE                 only its 32 single-rate instructions
  MAD / TRANS / CVT and the six pairs of {MAD, TRANS, CVT, SIMPLE}   single classes and class pairs: which pairs overlap
  REAL1 / REAL2          the production loops VERBATIM (real registers, real dependencies): the calibration that the harness
                         reproduces the production loop's own rate
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monte-carlo-simulation-of-hull-white-model-and-sensitivities-computation_amd")
OUT = os.path.join(ROOT, "tools", "mix_bound_body.inc")

TRANS = ("v_log_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32")
HALF = ("v_mad_u64_u32", "v_cvt_f32_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_cvt_f64_f32", "v_fma_f64", "v_fmac_f64",
        "v_add_f64", "v_mul_f64", "v_pk_fma_f32")


def klass(op):
    if op.startswith(TRANS):
        return "quarter"
    if op.startswith(HALF):
        return "half"
    return "full"


def listing():
    sys.path.insert(0, ROOT)
    import hwmc
    flags = [f for f in hwmc.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "eng.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *flags, "--cuda-device-only", "-S", "-o", out,
                               os.path.join(PKG, "csrc", "hw_engine.hip")], stderr=subprocess.DEVNULL)
        return open(out).read()


def hot_loops(text, pattern):
    m = re.search(r"^(_ZN3hwk[^\n:]*%s[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % re.escape(pattern), text, re.S | re.M)
    assert m, pattern
    lines = m.group(2).split("\n")
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
        ins = [l.strip() for l in lines[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
        if sum(x.startswith("v_mad_u64_u32") for x in ins) >= 10:
            out.append(ins)
    return out


def production_loop(text, pattern, mads):
    """The float (y, W) loop copy with `mads` wide multiplies: the first hot loop of the kernel that has exactly that many and
    no double-precision or exp instructions (the per-step fallback loop under weak mean reversion has both)."""
    for ins in hot_loops(text, pattern):
        ops = [x.split()[0] for x in ins]
        if sum(o == "v_mad_u64_u32" for o in ops) == mads and not any("f64" in o or o.startswith("v_exp") for o in ops):
            best = ins
            # prefer the copy without v_mov (the other float copy carries one per block for its loop-carried state)
            if not any(o.startswith("v_mov") for o in ops):
                return ins
    return best


REG = re.compile(r"\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b")


class Renamer:
    def __init__(self):
        self.next_dest = 8
        self.smap = {}

    def dest(self, width):
        if width == 2 and self.next_dest % 2:
            self.next_dest += 1
        if self.next_dest + width > 40:
            self.next_dest = 8
        r = self.next_dest
        self.next_dest += width
        return "v%d" % r if width == 1 else "v[%d:%d]" % (r, r + 1)

    @staticmethod
    def vsrc(n, width):
        base = (n % 4) + 4 * ((n // 4) % 2)
        if width == 2:
            base -= base % 2
            return "v[%d:%d]" % (base, base + 1)
        return "v%d" % base

    def ssrc(self, n, width=1):
        if n not in self.smap:
            self.smap[n] = 40 + len(self.smap)
            assert self.smap[n] < 100
        r = self.smap[n]
        return "s%d" % r if width == 1 else "s[%d:%d]" % (r, r + 1)


def strip_deps(ins, rn):
    """One production instruction -> its dependency-free twin (or the instruction itself for s_setprio / s_nop; None for
    the production's loop control)."""
    op = ins.split()[0]
    if op in ("s_setprio", "s_nop"):
        return ins
    if op.startswith(("s_cbranch", "s_cmp", "s_branch")):
        return None
    rest = ins[len(op):].strip()
    operands = [o.strip() for o in rest.split(",")]
    if op.startswith("s_"):                                   # scalar ALU: rename consistently, dependencies kept
        def sub(m):
            if m.group(1) == "s":
                a, b = int(m.group(2)), int(m.group(3))
                for k in range(a, b + 1):
                    rn.ssrc(k)
                return "s[%d:%d]" % (rn.smap[a], rn.smap[a] + (b - a))     # (pairs: allocated consecutively just above)
            return rn.ssrc(int(m.group(5)))
        return op + " " + ", ".join(REG.sub(sub, o) for o in operands)
    assert op.startswith("v_"), ins
    new = []
    for k, o in enumerate(operands):
        is_dest = k == 0
        is_sdst = op == "v_mad_u64_u32" and k == 1

        def sub(m, is_dest=is_dest, is_sdst=is_sdst):
            if m.group(1):                                     # register range
                cls, a, b = m.group(1), int(m.group(2)), int(m.group(3))
                if cls == "v":
                    return rn.dest(b - a + 1) if is_dest else rn.vsrc(a, b - a + 1)
                return "s[36:37]" if is_sdst else rn.ssrc(a, b - a + 1)
            cls, n = m.group(4), int(m.group(5))
            if cls == "v":
                return rn.dest(1) if is_dest else rn.vsrc(n, 1)
            return rn.ssrc(n)
        new.append(REG.sub(sub, o))
    return op + " " + ", ".join(new)


def regs_of(lines):
    v, s = set(), set()
    for l in lines:
        for m in REG.finditer(l[len(l.split()[0]):]):
            if m.group(1):
                (v if m.group(1) == "v" else s).update(range(int(m.group(2)), int(m.group(3)) + 1))
            else:
                (v if m.group(4) == "v" else s).add(int(m.group(5)))
    return v, s


def emit(name, lines, blocks, fh, census=None):
    v, s = regs_of(lines)
    v |= set(range(0, 8))
    s |= {36, 37}
    body = "".join('    "%s\\n\\t"\n' % l for l in lines)
    fh.write("// ---- %s: %d instructions per trip, %d Philox block(s) per trip\n" % (name, len(lines), blocks))
    if census:
        fh.write("//      VALU by issue class: %s\n" % census)
    fh.write("#define MIX_BLOCKS_%s %d\n" % (name, blocks))
    fh.write("#define MIX_BODY_%s \\\n%s\n" % (name, body.rstrip("\n").replace("\n", " \\\n")))
    clob = ['"v%d"' % k for k in sorted(v)] + ['"s%d"' % k for k in sorted(s)] + ['"vcc"', '"scc"', '"memory"']
    fh.write("#define MIX_CLOBBERS_%s %s\n\n" % (name, ", ".join(clob)))


def census_of(lines):
    c = {"full": 0, "half": 0, "quarter": 0}
    for l in lines:
        op = l.split()[0]
        if op.startswith("v_"):
            c[klass(op)] += 1
    return c


def generate(search=0):
    text = listing()
    p1 = production_loop(text, "zbc_superposed_kernelIfLi1ELi1E", 16)
    p2 = production_loop(text, "zbc_superposed_kernelIfLi1ELi2E", 32)
    import io
    fh = io.StringIO()
    fh.write("// GENERATED by tools/gen_mix_bound.py from the gfx950 listing of csrc/hw_engine.hip (production flags) -- do not edit.\n"
             "// tests/test_kernel_census.py regenerates it and fails if this file is stale.\n\n")
    variants = {}
    for name, loop, blocks in (("PROD1", p1, 1), ("PROD2", p2, 2)):
        rn = Renamer()
        free = [x for x in (strip_deps(i, rn) for i in loop) if x]
        variants[name] = (free, blocks)
        variants[name + "_NOPRIO"] = ([x for x in free if not x.startswith(("s_setprio", "s_nop"))], blocks)
        real = [i for i in loop if not i.startswith(("s_cbranch", "s_cmp", "s_branch"))]
        variants["REAL" + name[-1]] = (real, blocks)
        if name == "PROD1":      # the same loop on priority levels 2/3 instead of 0/1 (tools/pace_lab.hip: rotating priorities)
            variants["REAL1_HI"] = ([re.sub(r"^s_setprio ([01])$", lambda m: "s_setprio %d" % (int(m.group(1)) + 2), x) for x in real], blocks)
    valu = [x for x in variants["PROD1_NOPRIO"][0] if x.startswith("v_")]
    order = {"half": 0, "quarter": 1, "full": 2}
    grouped = sorted(valu, key=lambda x: (order[klass(x.split()[0])], x.split()[0].startswith("v_cvt")))
    variants["GROUPED"] = (grouped, 1)
    n_excl = sum(klass(x.split()[0]) != "full" for x in grouped)
    variants["GROUPED_PRIO"] = (["s_setprio 1"] + grouped[:n_excl] + ["s_setprio 0"] + grouped[n_excl:], 1)
    variants["EXCL"] = ([x for x in valu if klass(x.split()[0]) != "full"], 1)
    variants["SIMPLE"] = ([x for x in valu if klass(x.split()[0]) == "full"], 1)
    # which classes overlap with which (production relative order, no priorities): the co-issue matrix behind the bound
    is_mad = lambda x: x.startswith("v_mad_u64_u32")
    is_cvt = lambda x: x.startswith("v_cvt")
    is_trans = lambda x: klass(x.split()[0]) == "quarter"
    is_simple = lambda x: klass(x.split()[0]) == "full"
    for nm, pred in (("MAD", is_mad), ("TRANS", is_trans), ("CVT", is_cvt),
                     ("MAD_TRANS", lambda x: is_mad(x) or is_trans(x)), ("MAD_SIMPLE", lambda x: is_mad(x) or is_simple(x)),
                     ("TRANS_SIMPLE", lambda x: is_trans(x) or is_simple(x)), ("CVT_SIMPLE", lambda x: is_cvt(x) or is_simple(x)),
                     ("MAD_CVT", lambda x: is_mad(x) or is_cvt(x)), ("TRANS_CVT", lambda x: is_trans(x) or is_cvt(x))):
        variants[nm] = ([x for x in valu if pred(x)], 1)
    if search:
        return search_variants(valu, search, (is_mad, is_cvt, is_trans, is_simple))
    c1, c2 = census_of(variants["PROD1"][0]), census_of(variants["PROD2"][0])
    fh.write("#define MIX_CENSUS_PROD1_FULL %d\n#define MIX_CENSUS_PROD1_HALF %d\n#define MIX_CENSUS_PROD1_QUARTER %d\n"
             % (c1["full"], c1["half"], c1["quarter"]))
    fh.write("#define MIX_CENSUS_PROD2_FULL %d\n#define MIX_CENSUS_PROD2_HALF %d\n#define MIX_CENSUS_PROD2_QUARTER %d\n\n"
             % (c2["full"], c2["half"], c2["quarter"]))
    for name, (lines, blocks) in variants.items():
        emit(name, lines, blocks, fh, census_of(lines))
    full = {"PROD1", "PROD1_NOPRIO", "PROD2", "PROD2_NOPRIO", "GROUPED", "GROUPED_PRIO", "REAL1", "REAL2", "REAL1_HI"}
    fh.write("#define MIX_VARIANTS(X) \\\n" + " \\\n".join(
        "    X(%s, %d, %d)" % (name, 0 if name.startswith("REAL") else 1, 1 if name in full else 0) for name in variants if name != "REAL1_HI") + "\n")
    return fh.getvalue()


def search_variants(valu, n, preds):
    """n random ORDERINGS and priority schemes of the production loop's 60 dependency-free instructions (any order is valid
    once the dependencies are gone): the classes' instructions are dealt out in chunks of a random size per class, the
    classes taken in a random cyclic order, and a random subset of the classes is issued at priority 1 (s_setprio at every
    change, with or without the s_nop the production code puts behind it).  Fixed seed: the search is reproducible.  If one
    of them ran faster than the production ordering, the 'bound' would not be one."""
    import io
    import random
    rng = random.Random(20250605)
    names = ("mad", "cvt", "trans", "simple")
    pools = [[x for x in valu if p(x)] for p in preds]
    fh = io.StringIO()
    fh.write("// GENERATED by tools/gen_mix_bound.py --search %d (random orderings / priority schemes of PROD1's instructions) -- do not edit.\n\n" % n)
    fh.write("#define MIX_CENSUS_PROD1_FULL 32\n#define MIX_CENSUS_PROD1_HALF 20\n#define MIX_CENSUS_PROD1_QUARTER 8\n\n")
    listed = []
    for k in range(n):
        chunk = [rng.choice((1, 2, 4, 8, 16)), rng.choice((1, 2, 4)), rng.choice((1, 2, 4, 8)), rng.choice((1, 2, 4, 8, 16, 32))]
        order = list(range(4))
        rng.shuffle(order)
        hi = [rng.random() < 0.5 for _ in range(4)]
        nop = rng.random() < 0.5
        left = [list(p) for p in pools]
        seq = []
        while any(left):
            for c in order:
                for _ in range(chunk[c]):
                    if left[c]:
                        seq.append((c, left[c].pop(0)))
        lines, cur = [], None
        use_prio = any(hi) and not all(hi)
        for c, ins in seq:
            if use_prio and hi[c] != cur:
                lines.append("s_setprio %d" % (1 if hi[c] else 0))
                if nop:
                    lines.append("s_nop 0")
                cur = hi[c]
        # (the priority a trip ends on is the one the next trip starts on: a wave keeps it across the back edge)
            lines.append(ins)
        name = "S%03d" % k
        desc = "order %s chunks %s prio1 {%s}%s" % ("-".join(names[c] for c in order), "/".join("%s %d" % (names[c], chunk[c]) for c in range(4)),
                                                  ",".join(names[c] for c in range(4) if hi[c]) if use_prio else "none",
                                                  " +s_nop" if (use_prio and nop) else "")
        fh.write("// %s: %s\n" % (name, desc))
        emit(name, lines, 1, fh, census_of(lines))
        listed.append(name)
    fh.write("#define MIX_VARIANTS(X) \\\n" + " \\\n".join("    X(%s, 1, 1)" % nm for nm in listed) + "\n")
    return fh.getvalue()


def main():
    if "--search" in sys.argv:
        n = int(sys.argv[sys.argv.index("--search") + 1])
        out = os.path.join(ROOT, "tools", "mix_search_body.inc")
        with open(out, "w") as fh:
            fh.write(generate(search=n))
        print("wrote", out)
        return
    text = generate()
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else OUT     # --out FILE: write somewhere else (build() does, for a stale body)
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("mix_bound_body.inc is %s" % ("up to date" if ok else "STALE: run python tools/gen_mix_bound.py"))
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        fh.write(text)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
