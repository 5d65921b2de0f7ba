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
  F64_1 / F64_1_NOPRIO / REALF64_1   the same three for the DOUBLE ZBC loop, zbc_superposed_kernel<double,1,1> (family zbc_f64)
  CURVE1 / CURVE2 / CURVE2_F64       the loop of zcb_curve_blocked_kernel<1,float> on the 1000-step grid (BASELINE configs[0]: a maturity
                         every 2.5 Philox blocks) and of <2,float> / <2,double> on the 4000-step grid (configs[4]: every 10 blocks), maturity
                         events and the MaturityTree butterfly included (families curve_f32_p1, curve_f32_p2, curve_f64_p2).  That
                         loop has wave-uniform control flow (no event / event after the first half / after the second; 0-4 butterfly
                         levels; the flush of every sixteenth maturity), so its twin is built from TRACES: the scalar compares and
                         branches of the listing are resolved by a small scalar emulator (ScalarEmu) seeded with the registers the
                         MaturityTree asm names (event code, maturity index, n_mat), one trace per kind of trip; a twin trip is one
                         PERIOD of sixteen maturities -- every kind of trip as often as the production loop takes it in a period,
                         each kind in a counted scalar loop of its own (the order of the kinds differs from production, the multiset
                         is the same; the LDS accesses of the flush and the production's own compares/branches are left out, which can
                         only lower the figure).
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
HALF = ("v_mad_u64_u32", "v_cvt_f32_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_fma_f64", "v_fmac_f64",
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


def production_loop(text, pattern, mads, f64=False):
    """The (y, W) loop copy with `mads` wide multiplies: the first hot loop of the kernel that has exactly that many and
    no exp instructions; float kernels: no double-precision instructions either (the per-step fallback loop under weak mean
    reversion has both); f64: the double kernel's block-form loop (v_fma_f64 present)."""
    best = None
    for ins in hot_loops(text, pattern):
        ops = [x.split()[0] for x in ins]
        if sum(o == "v_mad_u64_u32" for o in ops) == mads and not any(o.startswith("v_exp") for o in ops) \
                and any("f64" in o for o in ops) == f64:
            best = ins
            # prefer the copy without v_mov (the other float copy carries one per block for its loop-carried state)
            if not any(o.startswith("v_mov") for o in ops):
                return ins
    return best


REG = re.compile(r"\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b")


class Renamer:
    def __init__(self, readonly=None):
        """readonly: production SGPRs the loop only READS (constants, lane masks).  They carry no dependency, so they share a
        small pool (s40..s47) instead of a scratch register each -- the curve loops name up to 38 SGPRs and the harness
        leaves 36.  None: every SGPR gets a scratch register of its own (the ZBC loops, as in round 3)."""
        self.next_dest = 8
        self.smap = {}
        self.readonly = readonly
        self.next_s = 40 if readonly is None else 48

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
        """scratch SGPR for production SGPR n (consistent within a variant); width 2: an even-aligned pair for s[n:n+1]"""
        if self.readonly is not None and all(k in self.readonly for k in range(n, n + width)):
            return "s%d" % (40 + n % 4) if width == 1 else "s[%d:%d]" % (44 + 2 * ((n // 2) % 2), 45 + 2 * ((n // 2) % 2))
        key = (n, width)
        if key not in self.smap:
            if width == 2:
                if (n, 1) in self.smap and (n + 1, 1) in self.smap and self.smap[(n, 1)] % 2 == 0 and self.smap[(n + 1, 1)] == self.smap[(n, 1)] + 1:
                    self.smap[key] = self.smap[(n, 1)]
                else:
                    self.next_s += self.next_s % 2
                    self.smap[key] = self.next_s
                    self.smap.setdefault((n, 1), self.next_s)
                    self.smap.setdefault((n + 1, 1), self.next_s + 1)
                    self.next_s += 2
            else:
                self.smap[key] = self.next_s
                self.next_s += 1
            assert self.next_s <= 72, "scratch SGPRs run into the harness's own (s72..s87)"
        r = self.smap[key]
        return "s%d" % r if width == 1 else "s[%d:%d]" % (r, r + 1)


MODIFIER = re.compile(r"\s(quad_perm:|row_shl:|row_shr:|row_ror:|row_mirror|row_half_mirror|row_bcast:|wave_|row_mask:|bank_mask:|bound_ctrl:|"
                      r"bitop3:|op_sel|neg_lo:|neg_hi:|clamp|mul:|div:|dst_sel:|src0_sel:|src1_sel:|dst_unused:)")


def split_operands(rest):
    """'v38, v6, v38 quad_perm:[1,0,3,2] row_mask:0xf' -> (['v38', 'v6', 'v38'], ' quad_perm:[1,0,3,2] row_mask:0xf')"""
    m = MODIFIER.search(rest)
    tail = rest[m.start():] if m else ""
    head = rest[:m.start()] if m else rest
    return [o.strip() for o in head.split(",")], tail


DROPPED = ("s_cbranch", "s_cmp", "s_branch", "s_bitcmp", "s_waitcnt", "ds_read", "ds_write", "ds_add")


def strip_deps(ins, rn):
    """One production instruction -> its dependency-free twin (or the instruction itself for s_setprio / s_nop; None for
    the production's loop control, and for the LDS accesses + waits of the curve kernel's flush)."""
    op = ins.split()[0]
    if op in ("s_setprio", "s_nop"):
        return ins
    if op.startswith(DROPPED):
        return None
    rest = ins[len(op):].strip()
    operands, tail = split_operands(rest)
    if op.startswith("s_"):                                   # scalar ALU: rename consistently, dependencies kept
        def sub(m):
            if m.group(1) == "s":
                a, b = int(m.group(2)), int(m.group(3))
                return rn.ssrc(a, b - a + 1)
            if m.group(4) == "s":
                return rn.ssrc(int(m.group(5)))
            return m.group(0)
        return op + " " + ", ".join(REG.sub(sub, o) for o in operands) + tail
    assert op.startswith("v_"), ins
    new = []
    for k, o in enumerate(operands):
        is_dest = k == 0
        is_sdst = (op == "v_mad_u64_u32" and k == 1) or (op.startswith("v_cmp") and k == 0)

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
    return op + " " + ", ".join(new) + tail


# ---------------------------------------------------------------------------------------------------------------------------
# Loops with wave-uniform control flow (the curve kernel): traces through the listing, scalar branches resolved by emulation
# ---------------------------------------------------------------------------------------------------------------------------
def kernel_lines(text, pattern):
    """Instructions and labels of one kernel of the listing, comments and directives removed (inline-asm lines kept)."""
    m = re.search(r"^(_ZN3hwk[^\n:]*%s[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel" % re.escape(pattern), text, re.S | re.M)
    assert m, pattern
    out = []
    for l in m.group(2).split("\n"):
        l = l.split(";")[0].strip() if not l.strip().startswith(";") else ""
        if l and not (l.startswith(".") and not re.match(r"^\.LBB\d+_\d+:", l)):
            out.append(l)
    return out


LABEL = re.compile(r"^(\.LBB\d+_\d+|\d+):$")


def natural_loops(lines):
    """(index of the loop's label, index of its last backward branch) for every label that is the target of a backward branch;
    innermost = the ones that contain no loop of another label."""
    pos = {}
    for i, l in enumerate(lines):
        m = LABEL.match(l)
        if m and m.group(1).startswith("."):
            pos[m.group(1)] = i
    last = {}
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)$", l)
        if m and m.group(1) in pos and pos[m.group(1)] <= i:
            last[m.group(1)] = max(last.get(m.group(1), -1), i)
    loops = [(pos[k], v) for k, v in last.items()]
    return sorted(a for a in loops if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in loops))


class Unknown(Exception):
    pass


class ScalarEmu:
    """Just enough of the scalar ALU to resolve the wave-uniform branches of the curve loop: values are Python ints (32-bit
    registers; a 64-bit operand is lo | hi << 32), None = unknown.  exec is all ones.  Anything not modelled makes its
    destination (and scc) unknown; branching on an unknown raises."""
    M32, M64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF

    def __init__(self):
        self.r, self.scc, self.vcc, self.v = {}, None, None, {}

    def get(self, tok, wide=False):
        tok = tok.strip()
        if tok == "exec":
            return self.M64
        if tok == "vcc":
            return self.vcc
        m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
        if m:
            lo, hi = self.r.get(int(m.group(1))), self.r.get(int(m.group(2)))
            return None if lo is None or hi is None else lo | (hi << 32)
        m = re.match(r"^s(\d+)$", tok)
        if m:
            return self.r.get(int(m.group(1)))
        try:
            v = int(tok, 0)
            return v & (self.M64 if wide else self.M32)
        except ValueError:
            return None

    def put(self, tok, v, wide=False):
        tok = tok.strip()
        if tok == "vcc":
            self.vcc = v
            return
        m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
        if m:
            a = int(m.group(1))
            self.r[a] = None if v is None else v & self.M32
            self.r[a + 1] = None if v is None else (v >> 32) & self.M32
            return
        m = re.match(r"^s(\d+)$", tok)
        if m:
            self.r[int(m.group(1))] = None if v is None else v & self.M32

    @staticmethod
    def s32(v):
        return v - (1 << 32) if v is not None and v & 0x80000000 else v

    def step(self, ins):
        op = ins.split()[0]
        ops, _ = split_operands(ins[len(op):].strip())
        if op.startswith("v_"):                                 # vector instructions that write scalar registers
            # (the compiler sometimes parks a wave-uniform flag in a VGPR: v_cndmask vD, imm, imm, <mask> ... v_cmp_ne s[..], imm, vD)
            def uni(tok):
                mm = re.match(r"^v(\d+)$", tok)
                if mm:
                    return self.v.get(int(mm.group(1)))
                try:
                    return int(tok, 0) & self.M32
                except ValueError:
                    return None
            cm = re.match(r"^v_cmp_(eq|ne|lg)_(u32|i32)_e(32|64)$", op)
            if cm:
                dst = ops[0] if cm.group(3) == "64" else "vcc"
                a, b = (uni(ops[1]), uni(ops[2])) if cm.group(3) == "64" else (uni(ops[-2]), uni(ops[-1]))
                val = None if a is None or b is None else (self.M64 if (a == b) == (cm.group(1) == "eq") else 0)
                self.put(dst, val, True)
                return
            if op.startswith("v_cmp"):
                self.put(ops[0] if op.endswith("_e64") else "vcc", None, True)
                return
            if op == "v_cndmask_b32_e64" and re.match(r"^v\d+$", ops[0]):
                a, b, msk = uni(ops[1]), uni(ops[2]), self.get(ops[3], True)
                imm = not ops[1].startswith("v") and not ops[2].startswith("v")
                self.v[int(ops[0][1:])] = (b if msk else a) if imm and msk in (0, self.M64) and a is not None and b is not None else None
                return
            for mm in REG.finditer(ops[0]):                      # any other vector result is not wave-uniform knowledge
                if mm.group(1) == "v":
                    for k in range(int(mm.group(2)), int(mm.group(3)) + 1):
                        self.v.pop(k, None)
                elif mm.group(4) == "v":
                    self.v.pop(int(mm.group(5)), None)
            if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) or "_co_" in op:
                self.put(ops[1], None, True)
            elif op.startswith(("v_readfirstlane", "v_readlane")):
                self.put(ops[0], None)
            return
        if not op.startswith("s_") or op in ("s_setprio", "s_nop", "s_waitcnt", "s_sleep") or op.startswith(("s_cbranch", "s_branch")):
            return
        g = self.get
        cmp = re.match(r"^s_cmp_(eq|lg|lt|le|gt|ge)_(i32|u32|u64)$", op)
        if cmp:
            wide = cmp.group(2) == "u64"
            a, b = g(ops[0], wide), g(ops[1], wide)
            if a is None or b is None:
                self.scc = None
                return
            if cmp.group(2) == "i32":
                a, b = self.s32(a), self.s32(b)
            self.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[cmp.group(1)])
            return
        if op in ("s_bitcmp1_b32", "s_bitcmp0_b32"):
            a, b = g(ops[0]), g(ops[1])
            self.scc = None if a is None or b is None else int(((a >> (b & 31)) & 1) == (1 if op == "s_bitcmp1_b32" else 0))
            return
        if op in ("s_cselect_b32", "s_cselect_b64"):
            wide = op.endswith("b64")
            self.put(ops[0], None if self.scc is None else g(ops[1] if self.scc else ops[2], wide), wide)
            return
        if op in ("s_mov_b32", "s_mov_b64"):
            self.put(ops[0], g(ops[1], op.endswith("b64")), op.endswith("b64"))
            return
        two = {"s_and": lambda a, b: a & b, "s_or": lambda a, b: a | b, "s_xor": lambda a, b: a ^ b, "s_andn2": lambda a, b: a & ~b,
               "s_orn2": lambda a, b: a | ~b}
        m = re.match(r"^(s_andn2|s_orn2|s_and|s_or|s_xor)_(b32|b64)$", op)
        if m:
            wide = m.group(2) == "b64"
            a, b = g(ops[1], wide), g(ops[2], wide)
            if a == 0 and m.group(1) in ("s_and", "s_andn2"):
                v = 0
            elif b == 0 and m.group(1) == "s_and":
                v = 0
            else:
                v = None if a is None or b is None else two[m.group(1)](a, b) & (self.M64 if wide else self.M32)
            self.put(ops[0], v, wide)
            self.scc = None if v is None else int(v != 0)
            return
        if op in ("s_add_i32", "s_add_u32", "s_sub_i32", "s_sub_u32", "s_addc_u32", "s_subb_u32"):
            a, b = g(ops[1]), g(ops[2])
            c = self.scc if op in ("s_addc_u32", "s_subb_u32") else 0
            if a is None or b is None or c is None:
                self.put(ops[0], None)
                self.scc = None
                return
            v = a + b + c if "add" in op else a - b - c
            self.put(ops[0], v & self.M32)
            if op.endswith("u32"):
                self.scc = int(v > self.M32 or v < 0)
            else:
                sa, sb = self.s32(a), self.s32(b)
                t = sa + sb if "add" in op else sa - sb
                self.scc = int(not (-(1 << 31) <= t < (1 << 31)))
            return
        if op in ("s_lshl_b32", "s_lshr_b32"):
            a, b = g(ops[1]), g(ops[2])
            v = None if a is None or b is None else ((a << (b & 31)) if op == "s_lshl_b32" else (a >> (b & 31))) & self.M32
            self.put(ops[0], v)
            self.scc = None if v is None else int(v != 0)
            return
        # not modelled: destination and scc unknown
        if ops and re.match(r"^(s\d+|s\[\d+:\d+\]|vcc)$", ops[0]):
            self.put(ops[0], None, True)
        self.scc = None

    def taken(self, ins):
        op = ins.split()[0]
        if op == "s_branch":
            return True
        cond = {"s_cbranch_scc0": (self.scc, 0), "s_cbranch_scc1": (self.scc, 1),
                "s_cbranch_vccz": (None if self.vcc is None else int(self.vcc != 0), 0),
                "s_cbranch_vccnz": (None if self.vcc is None else int(self.vcc != 0), 1),
                "s_cbranch_execz": (1, 0), "s_cbranch_execnz": (1, 1)}[op]
        if cond[0] is None:
            raise Unknown(ins)
        return cond[0] == cond[1]


def tree_registers(body):
    """The SGPRs the MaturityTree asm statement names: it begins  s_cmp_eq_u32 <ev>, 0 / s_cbranch_scc1 9f /
    s_cmp_ge_i32 <m>, <n_mat> / s_cbranch_scc1 9f / s_bitcmp1_b32 <k>, 0  (csrc/hw_kernels.hpp, MaturityTree::add)."""
    for i in range(len(body) - 4):
        a = re.match(r"^s_cmp_eq_u32 (s\d+), 0$", body[i])
        b = re.match(r"^s_cmp_ge_i32 (s\d+), (s\d+)$", body[i + 2])
        c = re.match(r"^s_bitcmp1_b32 (s\d+), 0$", body[i + 4])
        if a and b and c and body[i + 1] == "s_cbranch_scc1 9f" and body[i + 3] == "s_cbranch_scc1 9f":
            return {"ev": a.group(1), "m": b.group(1), "n_mat": b.group(2), "k": c.group(1)}
    raise AssertionError("MaturityTree asm not found in the loop")


def curve_loop(text, pattern):
    """The blocked curve kernel's hot loop (the copy with the scalar round-2 multiply: 16 wide multiplies per pair):
    (lines of the whole span from the latch label to the last back edge, index of the trip's first instruction)."""
    lines = kernel_lines(text, pattern)
    P = int(re.search(r"ILi(\d)E", pattern).group(1))
    for a, b in natural_loops(lines):
        span = lines[a:b + 1]
        if sum(x.startswith("v_mad_u64_u32") for x in span) == 16 * P and any(x.startswith("v_exp_f32") for x in span):
            # the span opens with the latch (scalar bookkeeping + the loop's exit test); the trip starts at the next label
            head = next(i for i in range(1, len(span)) if LABEL.match(span[i]) and span[i].startswith("."))
            assert all(x.startswith("s_") for x in span[1:head]), "the latch is expected to be scalar"
            return span, head
    raise AssertionError("curve loop not found: " + pattern)


def trace_trip(span, head, ev, m):
    """The instructions ONE trip of the curve loop executes when the trip's event code is `ev` (0 none, 1 a maturity after
    the first half block, 2 after the second) and the maturity it completes is number `m`: walk from the trip's first
    instruction, resolving every scalar branch by emulation, until control returns to the latch; the latch's own scalar
    instructions are appended (they run once per trip)."""
    regs = tree_registers(span)
    emu = ScalarEmu()
    # to_event: the register the event code is selected from (s_cselect_b32 <ev>, <to_event>, 0 behind s_cmp_lt_i32 <to_event>, 3)
    sel = [re.match(r"^s_cselect_b32 %s, (s\d+), 0$" % regs["ev"], x) for x in span]
    sel = [x for x in sel if x]
    assert sel, "event-code select not found"
    emu.put(sel[0].group(1), ev if ev else 9)
    emu.put(regs["n_mat"], 101)
    emu.put(regs["m"], m - 1 if ev else m)                    # the trip increments it when it has an event
    labels = {}
    for i, l in enumerate(span):
        mm = LABEL.match(l)
        if mm:
            labels.setdefault(mm.group(1), []).append(i)

    def target(i, name):
        if name.endswith(("f", "b")) and name[:-1].isdigit():
            cands = [j for j in labels.get(name[:-1], []) if (j > i if name.endswith("f") else j < i)]
            return min(cands) if name.endswith("f") else max(cands)
        return labels[name][0] if name in labels else None    # None: leaves the loop

    out, i, steps = [], head, 0
    while True:
        steps += 1
        assert steps < 20000, "trace does not return to the latch"
        if i == 0 or i >= len(span):                           # back at the latch label (or fell off the end = the back edge)
            break
        l = span[i]
        if LABEL.match(l):
            i += 1
            continue
        op = l.split()[0]
        if op.startswith(("s_cbranch", "s_branch")):
            out.append(l)
            t = target(i, l.split()[1])
            if emu.taken(l):
                assert t is not None, "the trace leaves the loop at: " + l
                i = t
            else:
                i += 1
            continue
        emu.step(l)
        out.append(l)
        i += 1
    return out + span[1:head]


def curve_period(n_steps, n_mat=101):
    """The trips of the blocked curve loop over one period of sixteen maturities, in production order, cut into UNITS (one
    cycle of the event pattern: the trips until the distance to the next maturity is back at half a stride -- 5 trips with
    two events on the 1000-step grid, 10 trips with one on the 4000-step grid).  A trip's kind is (ev, levels, flush): the
    event logic of zcb_curve_blocked_kernel restated (to_event / ev), levels = butterfly merges = trailing one bits of
    m & 15, flush = every sixteenth maturity.  Returns ([(unit, representative m of each of its trips), ...], trips)."""
    stride = n_steps // (n_mat - 1)
    half = stride >> 1
    units, unit, to_event, m, blocks = [], [], half, 0, 0
    while m < 16 or unit:
        ev = to_event if to_event <= 2 else 0
        blocks += 1
        if ev:
            m += 1
            k = m & 15
            levels = 0
            while levels < 4 and (k >> levels) & 1:
                levels += 1
            unit.append(((ev, levels, k == 15), m))
        else:
            unit.append(((0, 0, False), 5))
        to_event = half - (2 - ev) if ev else to_event - 2
        if to_event == half:
            units.append(tuple(unit))
            unit = []
    assert m == 16, "the period does not close on a unit boundary"
    return units, blocks


def curve_variant(text, pattern, n_steps):
    """Dependency-free twin of one 16-maturity period of the curve loop.  The period is a sequence of units (curve_period); units
    with the same kinds of trips are run in ONE counted scalar loop, as often as the period holds them, and a run of three or
    more event-free trips inside a unit is a counted loop of its own: every event trip keeps the event-free trips production
    puts around it (what the SIMD's other waves are likely to be issuing at that moment); only the order of the units differs."""
    from collections import OrderedDict
    span, head = curve_loop(text, pattern)
    P = int(re.search(r"ILi(\d)E", pattern).group(1))
    units, blocks = curve_period(n_steps)
    groups = OrderedDict()
    for u in units:
        key = tuple(k for k, _ in u)
        g = groups.setdefault(key, {"count": 0, "unit": u})
        g["count"] += 1
    written = set()
    for l in span:
        if LABEL.match(l):
            continue
        op = l.split()[0]
        ops, _ = split_operands(l[len(op):].strip())
        dests = [ops[0]] if op.startswith("s_") and not op.startswith(DROPPED + ("s_setprio", "s_nop")) else []
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) or "_co_" in op:
            dests.append(ops[1])
        if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
            dests.append(ops[0])
        for d in dests:
            for mm in REG.finditer(d):
                if mm.group(1) == "s":
                    written.update(range(int(mm.group(2)), int(mm.group(3)) + 1))
                elif mm.group(4) == "s":
                    written.add(int(mm.group(5)))
    _, all_s = regs_of([x for x in span if not LABEL.match(x)])
    rn = Renamer(readonly=all_s - written)
    lines, census, label = [], {"full": 0, "half": 0, "quarter": 0}, 10
    kinds = OrderedDict()

    def body(kind, m):
        trip = trace_trip(span, head, kind[0], m)
        return [x for x in (strip_deps(i, rn) for i in trip) if x]

    for key, g in groups.items():
        lines.append("s_mov_b32 s72, %d" % g["count"])
        outer = label
        lines.append("%d:" % outer)
        label += 1
        u, i = g["unit"], 0
        while i < len(u):
            j = i
            while j < len(u) and u[j][0] == u[i][0] and u[i][0][0] == 0:
                j += 1
            run = max(j - i, 1)
            free = body(*u[i])
            c = census_of(free)
            for kk in census:
                census[kk] += c[kk] * run * g["count"]
            kinds[u[i][0]] = kinds.get(u[i][0], 0) + run * g["count"]
            if run >= 3:
                lines += ["s_mov_b32 s73, %d" % run, "%d:" % label] + free + ["s_sub_u32 s73, s73, 1", "s_cmp_lg_u32 s73, 0", "s_cbranch_scc1 %db" % label]
                label += 1
            else:
                lines += free * run
            i += run
        lines += ["s_sub_u32 s72, s72, 1", "s_cmp_lg_u32 s72, 0", "s_cbranch_scc1 %db" % outer]
    return lines, blocks * P, census, kinds


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
    assert not s & ({38, 39} | set(range(76, 88))), "%s uses a register of the harness: %s" % (name, sorted(s & ({38, 39} | set(range(76, 88)))))
    v |= set(range(0, 8))
    s |= {36, 37}
    if any(re.match(r"^\d+:$", l) for l in lines):            # counted sections (curve twins): the loop counters
        s |= {72, 73}
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
        if name == "PROD1":      # the same loop on priority levels 2/3 instead of 0/1 (profiles/r03_pace_lab.txt: rotating priorities)
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
    # ---- the other two production loops: the double ZBC loop and the curve loop
    family = {name: "zbc_f32" for name in variants}
    meta = {}                                                    # name -> (check mask, window scale)
    pd = production_loop(text, "zbc_superposed_kernelIdLi1ELi1E", 16, f64=True)
    rn = Renamer()
    free = [x for x in (strip_deps(i, rn) for i in pd) if x]
    variants["F64_1"] = (free, 1)
    variants["F64_1_NOPRIO"] = ([x for x in free if not x.startswith(("s_setprio", "s_nop"))], 1)
    variants["REALF64_1"] = ([i for i in pd if not i.startswith(("s_cbranch", "s_cmp", "s_branch"))], 1)
    for nm in ("F64_1", "F64_1_NOPRIO", "REALF64_1"):
        family[nm] = "zbc_f64"
    curve_census = {}
    for nm, pattern, n_steps, fam in (("CURVE1", "zcb_curve_blocked_kernelILi1EfE", 1000, "curve_f32_p1"),
                                      ("CURVE2", "zcb_curve_blocked_kernelILi2EfE", 4000, "curve_f32_p2"),
                                      ("CURVE2_F64", "zcb_curve_blocked_kernelILi2EdE", 4000, "curve_f64_p2")):
        lines, blocks, census, groups = curve_variant(text, pattern, n_steps)
        variants[nm] = (lines, blocks)
        family[nm] = fam
        # one trip = a whole period (40 / 320 pair-blocks): the deadline is checked every trip, and the window is stretched so
        # that the overshoot of the last trip stays ~1 % of it
        meta[nm] = (0, 1 if blocks <= 64 else 6)
        curve_census[nm] = (census, blocks, groups)
    c1, c2 = census_of(variants["PROD1"][0]), census_of(variants["PROD2"][0])
    fh.write("#define MIX_CENSUS_PROD1_FULL %d\n#define MIX_CENSUS_PROD1_HALF %d\n#define MIX_CENSUS_PROD1_QUARTER %d\n"
             % (c1["full"], c1["half"], c1["quarter"]))
    fh.write("#define MIX_CENSUS_PROD2_FULL %d\n#define MIX_CENSUS_PROD2_HALF %d\n#define MIX_CENSUS_PROD2_QUARTER %d\n\n"
             % (c2["full"], c2["half"], c2["quarter"]))
    for nm, (census, blocks, groups) in curve_census.items():
        fh.write("// %s: one trip = one period of 16 maturities = %d pair-blocks; kinds of trips (event, butterfly levels, flush) x count: %s\n"
                 % (nm, blocks, ", ".join("(%d,%d,%d) x %d" % (k[0], k[1], int(k[2]), cnt) for k, cnt in groups.items())))
    fh.write("\n")
    for name, (lines, blocks) in variants.items():
        emit(name, lines, blocks, fh, census_of(lines) if name not in curve_census else curve_census[name][0])
    full = {"PROD1", "PROD1_NOPRIO", "PROD2", "PROD2_NOPRIO", "GROUPED", "GROUPED_PRIO", "REAL1", "REAL2", "REAL1_HI",
            "F64_1", "F64_1_NOPRIO", "REALF64_1", "CURVE1", "CURVE2", "CURVE2_F64"}
    # X(name, dependency-free, full mix, family, deadline-check mask (trips), window scale)
    fh.write("#define MIX_VARIANTS(X) \\\n" + " \\\n".join(
        "    X(%s, %d, %d, \"%s\", %d, %d)" % (name, 0 if name.startswith("REAL") else 1, 1 if name in full else 0, family[name],
                                               meta.get(name, (15, 1))[0], meta.get(name, (15, 1))[1])
        for name in variants if name != "REAL1_HI") + "\n")
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
    fh.write("#define MIX_VARIANTS(X) \\\n" + " \\\n".join("    X(%s, 1, 1, \"zbc_f32\", 15, 1)" % nm for nm in listed) + "\n")
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
