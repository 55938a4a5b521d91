"""CPU (hipcc cross-compiles gfx950): the two level-0 kernels of the default V-cycle keep their planes in flight. What is
checked is the generated ISA, because the source cannot show it: a register ring that the compiler rotates with moves, one
load consumed in the step that issues it, or one load under a condition inside the loop, and every plane step of the march
waits for a request of the same step (`s_waitcnt vmcnt(0..1)` in the loop) -- DESIGN.md 5, "hidden waits". The V-cycle's
time moved by 3.5-5 % when these went away; nothing else in the test suite would notice them coming back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", "gmg_fused3d.hip")  # the unit that instantiates the fused level-0 kernels


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa") / "gmg.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", SRC, "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    lines = out.read_text().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2te\w+:", l)]
    bodies = {}
    for k, (i, name) in enumerate(starts):
        j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        end = next((e for e in range(i, j) if lines[e].startswith(".Lfunc_end")), j)
        bodies[name] = lines[i:end]
    return bodies


def main_loop(body):
    """(first line, last line) of the march over the planes: of all loops the one with the most barriers in it -- every plane
    step has at least one. A loop = the backward branches into blocks the compiler's comments assign to one loop header (the
    header itself need not be the first block of the loop in the file)."""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if not m:
            continue
        h = re.search(r"Header=BB(\d+_\d+)", m.group(2))
        header = ".LBB" + h.group(1) if h else (m.group(1) if "Loop Header" in m.group(2) else None)
        labels[m.group(1)] = (i, header)
    loops = {}
    for i, l in enumerate(body):
        m = re.search(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)][0] < i and labels[m.group(2)][1]:
            t, header = labels[m.group(2)]
            s0, e0 = loops.get(header, (t, i))
            loops[header] = (min(s0, t), max(e0, i))
    assert loops, "no loop found"
    return max(loops.values(), key=lambda se: sum("s_barrier" in l for l in body[se[0]:se[1] + 1]))


def waits_in(body, s, e):
    return [int(m.group(1)) for l in body[s:e + 1] for m in [re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", l)] if m]


CASES = [
    # mangled-name fragment, smallest vmcnt allowed inside the march, why
    ("k_rbgs_zero_resid3dILi32ELb0ELb1ELb0ELi4ELi0E", 6, "pre-sweep, level 0: four planes of f in flight, three steps between request and use"),
    ("k_rbgs_zero_resid3dILi32ELb0ELb0ELb1ELi4ELi0E", 6, "pre-sweep with exported ghost terms (level 1)"),
    ("k_rbgs_resweep_prolong3dILi32ELi27ELb0ELb0E", 2, "post-sweep, three workgroups per CU: the newest pair of f loads stays in flight"),
    ("k_rbgs_resweep_prolong3dILi32ELi59ELb0ELb0E", 2, "post-sweep at 512^3: two workgroups per CU, four slots"),
    ("k_rbgs_resweep_prolong3dILi32ELi3ELb1ELb0E", 2, "post-sweep with exported ghost terms (level 1)"),
]


@pytest.mark.parametrize("frag,floor,why", CASES, ids=[c[0][:40] for c in CASES])
def test_march_never_waits_for_its_newest_loads(isa, frag, floor, why):
    name = next((n for n in isa if frag in n), None)
    assert name is not None, f"kernel {frag} is not instantiated any more"
    body = isa[name]
    s, e = main_loop(body)
    assert sum("s_barrier" in l for l in body[s:e + 1]) >= 2, "that is not the march over the planes"
    w = waits_in(body, s, e)
    assert w and min(w) >= floor, (why, sorted(set(w)))
    assert not any("scratch_" in l for l in body[s:e + 1]), "register spills inside the march"


NO_SPILL = [  # every instantiation a default-option cycle or te_bicgstab launches on 32^3 patches (round 6: two of them spilled --
    # the pre-sweep that forms te_bicgstab's pending p and exports ghost terms, and the one that reads AND exports them -- with their
    # reloads inside the march, in the in-order queue of the planes in flight; they run two workgroups per CU now)
    "k_rbgs_zero_resid3dILi32ELb0ELb1ELb0ELi4ELi0E", "k_rbgs_zero_resid3dILi32ELb0ELb1ELb0ELi4ELi1E", "k_rbgs_zero_resid3dILi32ELb0ELb1ELb0ELi4ELi2E",
    "k_rbgs_zero_resid3dILi32ELb0ELb0ELb0ELi4ELi0E", "k_rbgs_zero_resid3dILi32ELb0ELb0ELb0ELi4ELi1E", "k_rbgs_zero_resid3dILi32ELb0ELb0ELb0ELi4ELi2E",
    "k_rbgs_zero_resid3dILi32ELb0ELb0ELb1ELi4ELi0E", "k_rbgs_zero_resid3dILi32ELb0ELb1ELb1ELi4ELi0E", "k_rbgs_zero_resid3dILi32ELb1ELb0ELb0ELi4ELi0E",
    "k_rbgs_resweep_prolong3dILi32ELi59ELb0ELb0E", "k_rbgs_resweep_prolong3dILi32ELi59ELb0ELb1E", "k_rbgs_resweep_prolong3dILi32ELi19ELb0ELb0E",
    "k_rbgs_resweep_prolong3dILi32ELi27ELb0ELb0E", "k_rbgs_resweep_prolong3dILi32ELi3ELb1ELb0E",
]


@pytest.mark.parametrize("frag", NO_SPILL, ids=[f[7:48] for f in NO_SPILL])
def test_default_path_kernels_do_not_spill(isa, frag):
    name = next((n for n in isa if frag in n), None)
    assert name is not None, f"kernel {frag} is not instantiated any more"
    spills = [l.strip() for l in isa[name] if "scratch_" in l]
    assert not spills, (frag, len(spills), spills[:3])


# ---------------------------------------------------------------------------------------------------------------------------------
# k_ps_sym (the reference smoother's patch solve): a plane is consumed a plane's worth of matrix instructions behind its request.
# Round 5 found the instruction scheduler hoisting the first instructions of a plane's y transform -- butterflies that depend on
# nothing but the plane's loads -- to right behind the request: `s_waitcnt vmcnt(5)` six MFMAs after sixteen loads, a whole HBM
# latency exposed per patch (the zero-guess variant ran 8 % slower); vector-ALU scheduling fences in phase A stopped it.
PS_SRC = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", "gmg_patchsolve.hip")


@pytest.fixture(scope="module")
def ps_isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa_ps") / "ps.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", PS_SRC, "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    lines = out.read_text().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2te8k_ps_sym\w+:", l)]
    bodies = {}
    for i, name in starts:
        end = next(e for e in range(i, len(lines)) if lines[e].startswith(".Lfunc_end"))
        bodies[name] = lines[i:end]
    return bodies, out.read_text()


def events(body):
    ev = []
    for l in body:
        m = re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", l)
        if m:
            ev.append(("W", int(m.group(1))))
        elif "global_load" in l:
            ev.append(("L", 0))
        elif "scratch_load" in l or "scratch_store" in l:
            ev.append(("X", 0))
        elif "v_mfma" in l:
            ev.append(("m", 0))
        elif "s_barrier" in l:
            ev.append(("|", 0))
    return ev


@pytest.mark.parametrize("variant", ["ILb0ELb0E", "ILb0ELb1E"], ids=["zero-guess", "zero-guess-faces"])
def test_patch_solve_planes_are_not_consumed_right_behind_their_request(ps_isa, variant):
    bodies, text = ps_isa
    name = next(n for n in bodies if variant in n)
    ev = events(bodies[name])
    assert not any(k == "X" for k, _ in ev), "k_ps_sym spills: scratch traffic waits for every plane in flight (in-order memory pipeline)"
    m = re.search(re.escape(name) + r".*?\.vgpr_spill_count:\s+(\d+)", text, flags=re.S)
    assert m and int(m.group(1)) == 0
    # bursts of >= 12 plane loads (a matrix instruction or two may sit inside one); behind each: no wait that needs the burst's own
    # loads (vmcnt <= 8) within the next 12 matrix instructions
    i, bursts = 0, 0
    while i < len(ev):
        if ev[i][0] != "L":
            i += 1
            continue
        j, loads, inner = i, 0, 0
        while j < len(ev) and (ev[j][0] == "L" or (ev[j][0] == "m" and inner < 2 and j + 1 < len(ev) and ev[j + 1][0] == "L")):
            loads += ev[j][0] == "L"
            inner += ev[j][0] == "m"
            j += 1
        if loads >= 12:
            bursts += 1
            mf, k = 0, j
            while k < len(ev) and mf < 12 and ev[k][0] != "|":
                if ev[k][0] == "m":
                    mf += 1
                if ev[k][0] == "W":
                    assert ev[k][1] > 8, (name, "a wait for loads just requested", [e for e in ev[i:k + 1]][-24:])
                k += 1
        i = j
    assert bursts >= 4  # (the prologue's two planes, the two re-requests of phase A, the reciprocal tables, the prefetches)


# ---------------------------------------------------------------------------------------------------------------------------------
# The z-slab kernels of the small levels (round 6, profiles/r06_tail_stamps.txt): a launch there is ONE dependent chain per workgroup,
# and what the stamps found in front of the first plane request was five to ten round trips -- kernel arguments fetched by scalar loads
# in four or five batches (behind the early exit, behind the `order` branch, ...), face tables entry by entry under the face-kind tests,
# a load under a branch in the middle of the prologue's data requests. Kept out by: argsUpFront (one batch), Reg6 (tables once, in
# registers), ProlongSrc::cbase, addresses selected instead of loads branched around. The ISA shows whether they are still out.
SLAB_SRC = os.path.join(ROOT, "pressurepoissonsolver_amd", "csrc", "gmg_launch3d.hip")
SLABS = [  # mangled-name fragment, most `s_waitcnt vmcnt(0)` allowed before the first barrier (the `order` entry + the tables; None: not counted)
    ("k_rbgs3dILi32ELb1ELb0ELi8E", 2, "sweep from zero"),
    ("k_rbgs3dILi32ELb0ELb1ELi8E", 2, "sweep on u + P e"),
    ("k_rbgs3dILi32ELb0ELb0ELi8E", 2, "plain sweep"),
    ("k_stencil3dILi32ELi3ELi8ELi0E", None, "residual + restriction (the blocks of a parent on another rank are fetched under a branch of their own)"),
    ("k_stencil3dILi32ELi1ELi8ELi0E", 2, "residual"),
]


@pytest.fixture(scope="module")
def slab_isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    out = tmp_path_factory.mktemp("isa_slab") / "l3d.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-x", "hip", SLAB_SRC, "-o", str(out)],
                   check=True, capture_output=True, timeout=900)
    lines = out.read_text().split("\n")
    bodies = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN2te\d*k_(?:rbgs3d|stencil3d)\w+):", l)
        if m:
            end = next(e for e in range(i, len(lines)) if lines[e].startswith(".Lfunc_end"))
            bodies[m.group(1)] = [x for x in lines[i:end] if not x.lstrip().startswith(";")]
    return bodies


@pytest.mark.parametrize("frag,max_drains,what", SLABS, ids=[s[2].split(" (")[0] for s in SLABS])
def test_slab_kernels_fetch_arguments_once_and_tables_once(slab_isa, frag, max_drains, what):
    name = next((n for n in slab_isa if frag in n), None)
    assert name is not None, f"kernel {frag} is not instantiated any more"
    body = slab_isa[name]
    first_branch = next(i for i, l in enumerate(body) if re.search(r"\bs_c?branch", l))
    first_barrier = next(i for i, l in enumerate(body) if "s_barrier" in l)
    # kernel arguments come through s[0:1] (the kernarg segment pointer): every scalar load from it sits in front of the first branch
    late = [l.strip() for l in body[first_branch:] if re.search(r"s_load_dword\w*\s+s\S+,\s*s\[0:1\]", l)]
    assert not late, (what, "kernel arguments fetched behind a branch", late[:4])
    assert sum(bool(re.search(r"s_load_dword\w*\s+s\S+,\s*s\[0:1\]", l)) for l in body[:first_branch]) >= 4
    if max_drains is not None:
        drains = sum("vmcnt(0)" in l for l in body[:first_barrier])
        assert drains <= max_drains, (what, f"{drains} full drains of the memory pipeline before the first plane step")
    assert not any("scratch_" in l for l in body), "register spills"
