"""Structural checks on the gfx950 assembly of the Gauss-Newton level kernels (no GPU needed: hipcc cross-compiles).

They pin two properties that no CPU-side numerical test can see and that cost a hung GPU box once:

* the work loop (one pass per pair drawn from the queue) is a plain loop whose header starts with the workgroup
  barrier.  When the write-back block and the draw of the next pair were two adjacent `if (tid == 0)` blocks, one either
  side of the back edge, the compiler threaded them into a loop that reached this barrier with thread 0 parked outside
  it, and every workgroup spun forever on the pair it had just finished;
* register spills stay out of the pixel loops (the kernels sit at the 128-VGPR cap of 4 waves per SIMD);
* every workgroup barrier is preceded, in its own basic block, by `s_waitcnt lgkmcnt(0)`.  The compiler left that wait
  out in front of the work loop's barrier (it counts on LDS operations being ordered across waves); on the GPU about one
  wave in 10^5 then read the PREVIOUS pair's ticket from LDS and summed the wrong pair's pixels into the new pair's
  normal equations (one wrong pose per ~30 000, no error raised).  The kernels now say the wait explicitly.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "photoconsistency-visual-odometry_amd", "csrc")


@pytest.fixture(scope="module")
def kernels():
    subprocess.run(["make", "-s", "-C", CSRC, "isa"], check=True, capture_output=True)
    out = {}
    for name, least in (("gn_kernels.s", 18), ("gn_slide_kernel.s", 3), ("gn_bilinear_kernel.s", 6)):
        lines = open(os.path.join(CSRC, "build", name)).read().split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN9phovo_hip.*gn_(level|fused)_kernel.*:", l)]
        assert len(starts) >= least, f"{name}: expected every storage x variant instantiation of the level kernels"
        for a in starts:
            b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
            out[lines[a].split(":")[0]] = lines[a:b + 1]
    return out


def _first_instruction(body, i):
    i += 1
    while body[i].strip().startswith(";") or not body[i].strip():
        i += 1
    return body[i].strip()


def test_work_loop_head_is_the_barrier(kernels):
    """From the header of the outermost loop the code must fall straight into the workgroup barrier: nothing that
    branches or narrows the exec mask may come first (register spill moves and waits may)."""
    for name, body in kernels.items():
        heads = [i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l]
        assert len(heads) == 1, f"{name}: expected exactly one outermost loop (the work queue), found {len(heads)}"
        if "gn_level_kernelILi64E" in name:
            # one wave per workgroup: the compiler drops every workgroup barrier (a wave is in step with itself and its LDS
            # operations execute in order), so neither hazard this test is about exists; the loop shape is still checked
            assert not any(l.strip() == "s_barrier" for l in body), f"{name}: a barrier in a single-wave workgroup?"
            continue
        i, waited = heads[0] + 1, False
        while True:
            t = body[i].strip()
            i += 1
            if not t or t.startswith(";"):
                continue
            if t == "s_barrier":
                break
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                waited = True
                continue
            harmless = t.startswith(("s_waitcnt", "s_nop", "v_writelane_b32", "v_readlane_b32", "s_mov_b32", "v_mov_b32",
                                     "scratch_"))
            assert harmless, f"{name}: '{t}' between the work loop's header and its barrier"
        assert waited, f"{name}: no s_waitcnt lgkmcnt(0) in front of the work loop's barrier"


def _lds_settled_before(lines, pos, labels_at, branches_to, seen, depth=0, settles=None, offends=None, visited=None):
    """Walks back from line `pos` over EVERY path of the control-flow graph: None if each of them meets an
    `s_waitcnt lgkmcnt(0)` before any LDS instruction, else a description of the first offender.  (settles / offends:
    other predicates for the same walk; visited: a set shared by the whole walk -- a block seen once on any path is either
    settled or being judged further up, so it is not walked again and the depth is not limited.)"""
    settles = settles or (lambda t: t.startswith("s_waitcnt") and "lgkmcnt(0)" in t)
    offends = offends or (lambda t: t.startswith("ds_"))
    j = pos
    while j >= 0:
        t = lines[j].strip()
        if settles(t):
            return None
        if offends(t):
            return f"instruction '{t}' (line {j + 1}) reaches line {pos + 2} without the wait"
        if re.match(r"^_Z\w+:", t):
            return None                                    # the kernel's entry: nothing outstanding
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            label = m.group(1)
            if label in seen or (visited is not None and label in visited):
                return None                                # a loop closed: its other ways in decide
            seen = seen | {label}
            if visited is not None:
                visited.add(label)
            elif depth > 12:
                return f"gave up at {label}"
            preds = list(branches_to.get(label, []))
            k = j - 1                                      # falls through from the block laid out in front of it?
            while k >= 0 and (not lines[k].strip() or lines[k].strip().startswith((";", "."))) and \
                    not re.match(r"^(\.LBB\d+_\d+|_Z\w+):", lines[k].strip()):
                k -= 1
            if k >= 0 and not lines[k].strip().startswith(("s_branch", "s_endpgm", "s_setpc")):
                preds.append(k + 1)                        # (walk starts at k)
            for q in preds:
                bad = _lds_settled_before(lines, q - 1, labels_at, branches_to, seen, depth + 1, settles, offends, visited)
                if bad:
                    return bad
            return None
        j -= 1
    return None


def test_every_barrier_waits_for_lds_first():
    """No LDS instruction of a wave may still be in flight when the wave arrives at a workgroup barrier: on every path
    into an s_barrier the last LDS instruction is followed by an s_waitcnt lgkmcnt(0).  (The sliding-window kernel has
    barriers behind exec-masked blocks of pure arithmetic and whole phases of nothing but `s_barrier` in a scalar loop,
    so the check follows the control-flow graph instead of looking at the barrier's own block.)"""
    subprocess.run(["make", "-s", "-C", CSRC, "isa"], check=True, capture_output=True)
    total = 0
    for name in ("gn_kernels", "gn_bilinear_kernel", "gn_slide_kernel", "gn_wide_kernels", "pyramid_kernels", "warp_kernels"):
        lines = open(os.path.join(CSRC, "build", name + ".s")).read().split("\n")
        # labels are unique per file (.LBB<function>_<block>)
        labels_at, branches_to = {}, {}
        for i, l in enumerate(lines):
            t = l.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                labels_at[m.group(1)] = i
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m:
                branches_to.setdefault(m.group(1), []).append(i)
        for i, l in enumerate(lines):
            if l.strip() != "s_barrier":
                continue
            total += 1
            bad = _lds_settled_before(lines, i - 1, labels_at, branches_to, frozenset())
            assert bad is None, f"{name}.s line {i + 1}: {bad}"
    assert total >= 50, "expected the barriers of all level-kernel instantiations"


def test_the_barrier_check_sees_an_unsettled_path():
    """The checker itself, on hand-written listings: an LDS write on ONE of two ways into the barrier's block is found; the
    same listing with a wait behind that write passes; a barrier-only loop is judged by the block in front of it."""
    def run(text):
        lines = text.strip().split("\n")
        branches_to = {}
        for i, l in enumerate(lines):
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", l.strip())
            if m:
                branches_to.setdefault(m.group(1), []).append(i)
        at = next(i for i, l in enumerate(lines) if l.strip() == "s_barrier")
        return _lds_settled_before(lines, at - 1, {}, branches_to, frozenset())
    two_ways = """
_Zkernel:
 s_waitcnt lgkmcnt(0)
 s_cbranch_scc1 .LBB0_2
 ds_write_b32 v0, v1
 {wait}
.LBB0_2:
 v_add_f64 v[0:1], v[0:1], v[2:3]
 s_barrier
"""
    assert "ds_write_b32" in run(two_ways.format(wait=""))
    assert run(two_ways.format(wait="s_waitcnt lgkmcnt(0)")) is None
    waiting_loop = """
_Zkernel:
 ds_max_i32 v0, v1
 {wait}
 s_branch .LBB0_3
.LBB0_2:
 s_endpgm
.LBB0_3:
 s_add_i32 s0, s0, 1
 s_barrier
 s_cbranch_scc0 .LBB0_3
"""
    assert "ds_max_i32" in run(waiting_loop.format(wait=""))
    assert run(waiting_loop.format(wait="s_waitcnt vmcnt(0) lgkmcnt(0)")) is None


def test_two_draws_from_the_queue(kernels):
    for name, body in kernels.items():
        n = sum("global_atomic_add" in l for l in body)
        # two draw sites (before the loop, in the write-back block), each with ONE add in draw_pair (the queue heads sit a
        # cache line apart; a single queue is queue 0 of eight); gn_level_kernel can also take its pairs from the hand-over
        # list of the sliding-window kernel (a second add per site), which appends to that list (one add, in its write-back
        # block)
        if "gn_level_kernel_slide" in name:
            want = 2 * 1 + 1
        elif "gn_level_kernel_bilinear" in name or "gn_fused_kernel" in name:
            want = 2 * 1
        else:
            want = 2 * 2
        assert n == want, f"{name}: {n} atomic adds, expected {want}"


def test_no_scratch_in_innermost_loops(kernels):
    for name, body in kernels.items():
        inner = set()          # names of innermost-loop header blocks, e.g. "BB10_29"
        for i, l in enumerate(body):
            if "This Inner Loop Header" in l:
                j = i
                while not re.match(r"^\.LBB\d+_\d+:", body[j]):
                    j -= 1
                inner.add(body[j].split(":")[0][2:])
        assert inner, f"{name}: no innermost loops found"
        current = None         # innermost loop the block being read belongs to, if any
        scratch, draws = {}, set()
        for l in body:
            block = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
            if block:
                m = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                own = l.split(":")[0][2:] if l.startswith(".LBB") else None
                current = own if own in inner else (m.group(1) if m and m.group(1) in inner else None)
            elif current and "scratch_" in l:
                scratch.setdefault(current, l.strip())
            elif current and "global_atomic_add" in l:
                draws.add(current)
        # (the loop in which a workgroup draws its next pair from the queues runs once per PAIR, not per pixel: a reload there
        # is a few cycles per alignment)
        for loop, l in scratch.items():
            assert loop in draws, f"{name}: scratch traffic inside innermost loop {loop}: {l}"


def test_lds_bound_loads_are_waited_for_completely_and_never_use_the_instruction_offset():
    """gn_level_kernel_bilinear_dma lands its taps in LDS with buffer_load ... lds.  Two things tools/probes/lds_dma_probe.hip
    measured on gfx950 are pinned in the assembly: (1) such a load carries no instruction offset (that field moves the LDS
    address along with the memory address); (2) on EVERY path from such a load to a read of LDS there is an
    `s_waitcnt vmcnt(0)` -- a count that lets the youngest loads stay out is unsound when register-bound loads have been
    issued behind the LDS-bound ones (they may retire first), so the kernel waits for everything, explicitly."""
    subprocess.run(["make", "-s", "-C", CSRC, "isa"], check=True, capture_output=True)
    lines = open(os.path.join(CSRC, "build", "gn_bilinear_kernel.s")).read().split("\n")
    labels_at, branches_to = {}, {}
    for i, l in enumerate(lines):
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels_at[m.group(1)] = i
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m:
            branches_to.setdefault(m.group(1), []).append(i)
    dma = [i for i, l in enumerate(lines) if re.match(r"^buffer_load_\w+ .* lds\b", l.strip())]
    assert len(dma) >= 4 * (6 + 12), "expected the LDS-bound tap loads of the four instantiations"       # fp64: 6 per chunk, fp32: 12
    for i in dma:
        assert "offset:" not in lines[i], f"line {i + 1}: an LDS-bound load with an instruction offset: {lines[i].strip()}"
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN9phovo_hip.*gn_level_kernel_bilinear_dma.*:", l)]
    assert len(starts) == 4
    reads = 0
    for a in starts:
        b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
        for i in range(a, b):
            # (the taps come back as 16-byte words -- fp64 planes, a row's pair -- or 4-byte words; the 8-byte reads are the pose
            # constants the pixel loop fetches from the fixed block where it uses them: no load ever lands there)
            if not re.match(r"^ds_read(_b128|_b32|2\w*_b32)\b", lines[i].strip()):
                continue
            reads += 1
            bad = _lds_settled_before(lines, i - 1, labels_at, branches_to, frozenset(),
                                      settles=lambda t: t.startswith("s_waitcnt") and "vmcnt(0)" in t,
                                      offends=lambda t: bool(re.match(r"^buffer_load_\w+ .* lds\b", t)), visited=set())
            assert bad is None, f"gn_bilinear_kernel.s line {i + 1}: {bad}"
    assert reads >= 4 * 12
    # and the walk with these predicates does see a partial wait
    listing = """
_Zkernel:
 buffer_load_dword v1, s[4:7], s8 offen lds
 buffer_load_dword v2, v1, s[4:7], 0 offen
 s_waitcnt vmcnt({n})
 ds_read_b32 v0, v3
""".strip().split("\n")
    for n, ok in ((1, False), (0, True)):
        got = _lds_settled_before([l.format(n=n) for l in listing], 3, {}, {}, frozenset(),
                                  settles=lambda t: t.startswith("s_waitcnt") and "vmcnt(0)" in t,
                                  offends=lambda t: bool(re.match(r"^buffer_load_\w+ .* lds\b", t)), visited=set())
        assert (got is None) == ok
