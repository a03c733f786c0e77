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
    for name, least in (("gn_kernels.s", 15), ("gn_slide_kernel.s", 3)):
        lines = open(os.path.join(CSRC, "build", name)).read().split("\n")
        starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN9phovo_hip.*gn_level_kernel.*:", l)]
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
        # The variant with the owner map in HBM (...Lb0ELb0ELb0E...) can run as the follow-up of the sliding-window kernel
        # and then draws until it finds a pair marked for it: thread 0's first draw is a small loop of its own IN FRONT
        # of the work loop (no barrier inside: only thread 0 is in it, and the queue heads only grow, so it ends).
        if len(heads) == 2 and "ELb0ELb0ELb0E" in name:
            assert not any(l.strip() == "s_barrier" for l in body[heads[0]:heads[1]]), f"{name}: barrier inside the draw loop"
            heads = heads[1:]
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


def test_every_barrier_waits_for_lds_first():
    subprocess.run(["make", "-s", "-C", CSRC, "isa"], check=True, capture_output=True)
    total = 0
    for name in ("gn_kernels", "gn_slide_kernel", "gn_wide_kernels", "pyramid_kernels", "warp_kernels"):
        lines = open(os.path.join(CSRC, "build", name + ".s")).read().split("\n")
        for i, l in enumerate(lines):
            if l.strip() != "s_barrier":
                continue
            total += 1
            j, verdict = i - 1, None
            while j >= 0 and verdict is None:
                t = lines[j].strip()
                if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                    verdict = "ok"
                elif t.startswith("ds_"):
                    verdict = f"LDS instruction '{t}' between the last wait and the barrier"
                elif re.match(r"^(\.LBB\d+_\d+|_Z\w+):", t):
                    verdict = "no s_waitcnt lgkmcnt(0) in the barrier's basic block"
                j -= 1
            assert verdict == "ok", f"{name}.s line {i + 1}: {verdict}"
    assert total >= 50, "expected the barriers of all level-kernel instantiations"


def test_two_draws_from_the_queue(kernels):
    for name, body in kernels.items():
        n = sum("global_atomic_add" in l for l in body)
        # Two draw sites.  The sliding-window and the bilinear kernel draw before the loop and in the write-back block, each
        # site with the single-queue add and the per-XCD-queue add of draw_pair; the sliding-window kernel can also append
        # to a hand-over list (one more add).  gn_level_kernel draws ONE PAIR AHEAD (gn_device.hpp, look-ahead): before
        # the loop and in the serial section of every pair's first iteration, each site with the ticket's add (draw_begin)
        # and the add of the walk over the other queues (draw_end), plus the hand-over append of the write-back block.
        if "gn_level_kernel_slide" in name:
            want = 2 * 2 + 1
        elif "gn_level_kernel_bilinear" in name:
            want = 2 * 2
        else:
            want = 2 * 2 + 1
        assert n == want, f"{name}: {n} atomic adds, expected {want}"


def _innermost_loops(body):
    """[(header index, [instruction lines])] of the innermost loops of one kernel, in program order."""
    out = []
    for i, l in enumerate(body):
        if "This Inner Loop Header" not in l:
            continue
        j = i
        while not re.match(r"^\.LBB\d+_\d+:", body[j]):
            j -= 1
        label = body[j].split(":")[0][2:]
        depth = re.search(r"Depth=(\d+)", l).group(1)
        tag = f"Header={label} Depth={depth}"
        ins, inside, k = [], True, j + 1
        while k < len(body) and k - j < 2500:
            t = body[k]
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", t):
                inside = tag in t
            elif inside and t.strip() and not t.strip().startswith((";", ".")):
                ins.append(t.strip())
            k += 1
        out.append((j, ins))
    return out


def test_pass_1_keeps_its_ring_of_depth_loads_in_flight(kernels):
    """Pass 1 of gn_level_kernel (owner map in LDS) walks a ring of four depth registers: four loads per trip, each
    refilled behind the last use of its value, and every wait inside the loop leaves three of them outstanding.  A
    spill reload in front of the loop, a register copy at its back edge or an LDS-DMA in flight all turn those waits into
    `vmcnt(0)` -- the ring then runs empty once per trip and the first (HBM-cold) iteration of every pair crawls."""
    seen = 0
    for name, body in kernels.items():
        if "gn_level_kernel_slide" in name or "gn_level_kernel_bilinear" in name or "ELb0ELb0ELb0E" in name:
            continue
        seen += 1
        assert not any("global_load_lds" in l for l in body), f"{name}: an LDS-DMA drains every later wait"
        head = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
        loops = [(j, ins) for j, ins in _innermost_loops(body) if j > head]
        j, pass1 = next((j, ins) for j, ins in loops if any(l.startswith("ds_max_i32") for l in ins))    # the scatter
        loads = [l for l in pass1 if l.startswith("buffer_load")]
        assert len(loads) == 4, f"{name}: pass 1 should refill four ring slots per trip, found {len(loads)}"
        waits = [l for l in pass1 if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert waits and not any("vmcnt(0)" in w for w in waits), f"{name}: pass 1 drains its ring: {waits}"
        if "ELb0ELb1ELb1EddE" in name:                 # the reference-exact throughput instantiations: exactly the ring's depth
            assert all("vmcnt(3)" in w for w in waits), f"{name}: pass 1 waits {waits}"
    assert seen >= 15


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
        for l in body:
            block = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)
            if block:
                m = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                own = l.split(":")[0][2:] if l.startswith(".LBB") else None
                current = own if own in inner else (m.group(1) if m and m.group(1) in inner else None)
            elif current and "scratch_" in l:
                raise AssertionError(f"{name}: scratch traffic inside innermost loop {current}: {l.strip()}")
