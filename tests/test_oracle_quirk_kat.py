"""Known-answer tests for the quirks of the reference's traversal that SURVEY.md 8a lists as "semantics to preserve" and
that tests/test_oracle_kat.py did not reach: the edge-padding rule (VoxelRT/VolumeRaytracer.cu:216-232,240), the axis
tie-break on exact ties (:293-313), the int-truncated inclusive region check that does not count the step (:325-341), the
head-only maxSteps test with the 8-step bounce rays (:386, Renderer.cu:141), the double-rounded world-entry box (:373-376),
the previous_cell break (:402-407) and the NextCell snap with its <,< / else order (:470-487).

Every case in tests/quirk_cases.py was stepped through the reference's text by hand; the derivations are the docstrings
below.  Each is checked against BOTH restatements (the C oracle and the event-tracing Python one, oracle/ref_py.py), and the
events prove that the ray reaches the lines the case is named for.  The same cases run on the GPU in
tests/test_gpu_parity.py::test_quirk_cases_on_gpu.  The reference itself holds no vectors and cannot be built here: parity
with it stays "partial" by construction.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import ref_py
from tests import helpers, quirk_cases

f32 = np.float32


def _check(vxo, name):
    case = quirk_cases.all_cases()[name]
    w, o, d = quirk_cases.build_case(vxo, case)
    e = case["expect"]
    pw = ref_py.PyWorld(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
    py = ref_py.raytrace(pw, o, d, case["max_steps"])
    cc = w.raytrace(o, d, case["max_steps"])
    for r, stats in ((py, py["stats"]), (cc, tuple(cc["stats"]))):
        assert bool(r["hit"]) == e["hit"] and r["steps"] == e["steps"] and stats == tuple(e["stats"]), name
        assert [float(x) for x in r["normal"]] == [float(x) for x in e["normal"]], name
        if e["hit"]:
            assert tuple(r["voxel"]) == tuple(e["voxel"]), name
            assert [float(x) for x in r["pos"]] == [float(f32(x)) for x in e["pos"]], name
    missing = [ev for ev in case["events"] if ev not in py["events"]]
    assert not missing, (name, missing, py["events"])
    return py


def test_soft_max_steps(vxo):
    """maxSteps is tested only at the head of the two-level loop (:386); each inner walk runs to its own end.
    overrun: tests/test_oracle_kat.py's brick-miss-then-hit ray with maxSteps = 8 (the bounce rays' budget,
      Renderer.cu:141): after the first round total = 2 + 3 = 5 < 8, so a whole second round runs: +0 coarse, +5 brick
      crossings -> a HIT reported with 10 steps, more than the budget.
    cut: the first brick six coarse cells away: coarse walk 6 crossings (cells 0..5 empty; cell 6 occupied: slab entry at
      t = 6.5625 -> point 6.625, the exit iteration is not counted), brick walk from local (5, 3.5, 3.5): probes x = 5, 6, 7,
      crossings onto 6, 7, 8 (8 <= 8: inclusive, counted) -> 3; total 9 >= 8 ends the loop with NO hit although voxel
      (61,11,11) lies straight ahead; with MAX_STEPS the same ray restarts at coarse 7.0 (int(7.0) != 6: no nudge), hits
      cell 7's box at step 0 (point stays the start, :266-269), walks 5 crossings to local x = 5 and hits: 14 steps."""
    _check(vxo, "soft_max_steps_overrun")
    _check(vxo, "soft_max_steps_cut")
    _check(vxo, "soft_max_steps_cut_unbounded")


def test_world_entry_box_is_rounded_once(vxo):
    """:373-376 builds the entry box from `dimensions - FLT_EPS_DDA` with FLT_EPS_DDA a DOUBLE (VolumeRaytracer.cuh:20), so the
    difference is rounded to float once: 64 - 1e-6 -> 64.0f (half an ulp below 64 is 1.9e-6), 16 - 1e-6 -> 15.99999905f.
    C = 64: a ray from x = 600 going -x (coarse 75) enters at t = 11, x = 64.0: int(64.0) == dim -> edge rule with padding on x
      (:216-232); step 0 probes the CLAMPED cell 63; its box [63.875, 64] x [1.375, 1.5]^2 is touched at t_min = -0.0
      (fminf(0.125, -0.0)), accepted because -0.0 < 0 is false (:148); hit at step 0 -> the point stays the start.  Brick-local
      start = 512 - 504 = 8.0 == f: edge rule again, clamped voxel (7,3,3) solid at step 0 -> coarse normal (:496-499); total
      steps 0 -> position = start * f = (512, 11.5, 11.5), normal = the ENTRY normal (-1,0,0) (:518-522).
    C = 16: the far face is 16 - 2^-20; t_min = 9 + 2^-20 exactly, entry x = 16 - 2^-20 -> cell 15, no edge rule; the same
      chain gives position x = 8 * (16 - 2^-20) = 128 - 2^-17, the float just below 128."""
    _check(vxo, "world_entry_dim64_edge_rule")
    _check(vxo, "world_entry_dim16_no_edge")


def test_edge_slide_previous_cell_hole(vxo):
    """A ray sliding down the far face of the world: origin (64, 20.5, 11.5), direction (-1e-30, -1, 0) (its squared x
    underflows, so normalize leaves it as it is).  Coarse start (8.0, 2.5625, 1.4375) is "outside" (8.0 < 8 fails) but misses
    the entry box (t_min = 9.5e23 > t_max = 2.56), so the walk starts in cell 8 == dim: padding on x and y (:216-232).
    Coarse walk: step 0 probes the clamped cell (7,2,1) (empty); tMax_x = (8 - 8.0)/-1e-30 = -0.0 is the smallest -> x step
    to cell 7, counted; step 1 probes (7,2,1) AGAIN; y step, counted; step 2: (7,1,1) is occupied, its box
    [7.875, 8] x [1, 1.75] x [1.25, 1.625] is entered through the y face at t = 0.8125 -> point (8, 1.75, 1.4375); 2 steps.
    Brick-local start (8, 6, 3.5): cell 8 == f, padded; tMax_x = tMax_y = -0.0: `x < y` fails, `y <= x` holds -> the TIE
    goes to y (:293-313): crossing (8, 6, 3.5) counted, then the x step (8, 6.0, 3.5) counted, then six more y crossings down
    to y = 0 (inclusive bound), 8 steps, 8 probes of column (7, ., 3) -- (7,5,3) twice; out of range at y = -1.
    Restart point (8.0, 1.0, 1.4375): int(8.0) = 8 != HitCell.x = 7, so `projectedCellIsSame` is false and NOTHING is nudged
    (:445-447).  The second coarse walk starts in the clamped cell (7,1,1) again: box touched with t_min = t_max = -0.0, accepted
    (-0.0 < 0.0 is false) -> coarse hit on previous_cell -> break (:402-407): a hole.  10 steps, no hit."""
    py = _check(vxo, "edge_slide_previous_cell_hole")
    assert "world_entry" not in py["events"] and "ulp_nudge" not in py["events"]


def test_next_cell_snap_order(vxo):
    """Leaving the world through a face at coordinate 0.  Vertical ray (0,-1,0) above brick (2,0,1) whose voxels miss its
    column: coarse 2 crossings + box entry at y = 0.5, brick walk from local (5.5, 4, 3.5): 5 crossings down to y = 0
    (0 >= 0: counted), out of range at -1.  Restart point (2.6875, 0.0, 1.4375) truncates to HitCell (2,0,1): all three
    components move one ulp along the ray (zero direction components toward +inf, :452-460) -> y = -1.4e-45, which STILL
    truncates to 0, so the cell is the same and one axis is snapped to NextCell = (2,-1,1) (:470-487):
    |diff| = (0.6875.., 1, 0.4375..) -> x is not strictly smallest, y is not -> z: start.z = 1.0.  From there the box is missed
    (its z range starts at 1.25 and d.z = 0), one more coarse crossing leaves the grid: 8 steps, no hit.
    snap_x: the same with the column at x fraction 0.1875 < 0.4375.  snap_y / tie: a -x ray leaving through x = 0 with
    |diff| = (1, 0.4375, 0.46875) -> y; with |diff.y| == |diff.z| neither strict test holds and the ELSE branch takes z."""
    assert "snap_z" in _check(vxo, "floor_exit_snap_z")["events"]
    assert "snap_x" in _check(vxo, "floor_exit_snap_x")["events"]
    assert "snap_y" in _check(vxo, "wall_exit_snap_y")["events"]
    ev = _check(vxo, "wall_exit_snap_tie_goes_to_z")["events"]
    assert "snap_z" in ev and "snap_y" not in ev


def test_searched_ray_with_region_check_inside_the_world(vxo):
    _check(vxo, "diagonal_ties_region_oob_nudge_snap_edge")


# ---- single-level DDARayTraversal (dense mode, no per-cell boxes) ---------------------------------------------------------
def _dda(vxo, voxels, start, d, dims=(16, 16, 16), bounds=None):
    v = np.zeros(dims, bool)
    for p in voxels:
        v[p] = True
    words = vxo.dense_from_voxels(v)
    P, R = vxo.DDAParams(), vxo.DDAResult()
    P.bits = words.ctypes.data_as(C.POINTER(C.c_uint32))
    P.nbits = dims[0] * dims[1] * dims[2]
    P.dims = (C.c_int * 3)(*dims)
    P.start = (C.c_float * 3)(*start)
    P.dir = (C.c_float * 3)(*d)
    P.max_steps = 2048
    if bounds is not None:
        P.has_bounds = 1
        P.bounds_min = (C.c_float * 3)(*bounds[0])
        P.bounds_max = (C.c_float * 3)(*bounds[1])
    vxo.lib().vxo_dda(C.byref(P), C.byref(R))
    ev = []
    py = ref_py.dda(words, dims, [f32(x) for x in start], [f32(x) for x in d], ev,
                    bounds=None if bounds is None else ([f32(x) for x in bounds[0]], [f32(x) for x in bounds[1]]))
    assert (bool(R.hit), bool(R.out_of_bounds), R.steps, R.probes) == (py.hit, py.oob, py.steps, py.probes)
    assert [float(x) for x in R.point] == [float(x) for x in py.point]
    return R, ev


def test_edge_rule_single_level(vxo):
    """:216-232,240-244.  Start x = 16.0 on a 16-cell axis, direction -x: cell 16 == dim -> padded range; lookups use the
    clamped cell 15, which is therefore probed TWICE (at cell 16 and at cell 15); crossings onto 16, 15, 14 are counted, voxel
    (13,3,3) ends the walk: 3 steps, 4 probes, point (14, 3.5, 3.5), NextCell (12,3,3).  With the voxel AT (15,3,3) the very
    first (clamped) probe hits: 0 steps, HitCell (15,3,3), point = start.  Direction +x from the same start gets no padding
    (`dx < 0` fails): out of range at step 0."""
    R, ev = _dda(vxo, [(13, 3, 3)], (16.0, 3.5, 3.5), (-1, 0, 0))
    assert (R.hit, R.out_of_bounds, R.steps, R.probes) == (1, 0, 3, 4) and "edge_pad" in ev and "clamped_lookup" in ev
    assert list(R.point) == [14.0, 3.5, 3.5] and list(R.hit_cell) == [13, 3, 3] and list(R.next_cell) == [12, 3, 3]
    assert list(R.normal) == [-1, 0, 0]
    R, ev = _dda(vxo, [(15, 3, 3)], (16.0, 3.5, 3.5), (-1, 0, 0))
    assert (R.hit, R.steps, R.probes) == (1, 0, 1) and list(R.hit_cell) == [15, 3, 3] and list(R.point) == [16.0, 3.5, 3.5]
    assert list(R.next_cell) == [15, 3, 3]
    R, ev = _dda(vxo, [(15, 3, 3)], (16.0, 3.5, 3.5), (1, 0, 0))
    assert (R.hit, R.out_of_bounds, R.steps, R.probes) == (0, 1, 0, 0) and "edge" in ev and "edge_pad" not in ev


def test_axis_tie_break_single_level(vxo):
    """:293-313: x only if STRICTLY smaller than both; else y if `y <= x` and strictly smaller than z; else z.
    Direction (1,1,1) from (0.5,0.5,0.5): all three equal -> z, then y (x == y, both < z), then x: the walk visits (0,0,1) and
    (0,1,1) and finds a voxel there after 2 steps with normal (0,1,0); under any other order it would never see that cell.
    Direction (1,1,0): z is infinite, x == y -> y first: voxel (0,1,0) is hit after 1 step."""
    s = float(f32(1.0) / np.sqrt(f32(3.0)))
    R, ev = _dda(vxo, [(0, 1, 1)], (0.5, 0.5, 0.5), (s, s, s))
    assert (R.hit, R.steps, R.probes) == (1, 2, 3) and list(R.hit_cell) == [0, 1, 1] and list(R.normal) == [0, 1, 0] and "tie" in ev
    h = float(f32(1.0) / np.sqrt(f32(2.0)))
    R, ev = _dda(vxo, [(0, 1, 0)], (0.5, 0.5, 0.5), (h, h, 0))
    assert (R.hit, R.steps, R.probes) == (1, 1, 2) and list(R.hit_cell) == [0, 1, 0] and list(R.normal) == [0, 1, 0]
    R, ev = _dda(vxo, [(1, 0, 0)], (0.5, 0.5, 0.5), (h, h, 0))  # the x neighbour is never visited before y
    assert list(R.hit_cell) != [1, 0, 0] or R.steps > 1


def test_region_check_single_level(vxo):
    """:325-341: bounds are truncated to int (3.9 -> 3) and INCLUSIVE; the crossing point that fails the check is neither
    counted nor stored.  +x ray from (0.5,0.5,0.5), region x in [0, 3.9]: crossings onto 1, 2, 3 are counted (3 <= 3), the
    crossing onto 4 fails -> out of bounds with 3 steps, 4 probes, point (3, 0.5, 0.5)."""
    R, ev = _dda(vxo, [], (0.5, 0.5, 0.5), (1, 0, 0), bounds=((0, 0, 0), (3.9, 16, 16)))
    assert (R.hit, R.out_of_bounds, R.steps, R.probes) == (0, 1, 3, 4) and "region_oob" in ev
    assert list(R.point) == [3.0, 0.5, 0.5]


def test_two_restatements_agree_and_cover_every_quirk(vxo):
    """The C oracle and the Python restatement, written separately from the reference's text, on 3000 adversarial rays
    (tests/helpers.mixed_rays) through two random worlds: identical results, and between them the rays reach every quirk."""
    seen = set()
    for wi, dens in enumerate((0.02, 0.15)):
        v = np.random.default_rng(wi).random((64, 64, 64)) < dens
        w = vxo.World.from_voxels(v, 8)
        pw = ref_py.PyWorld(w.factor, w.cdims, w.coarse_bits, w.brick_slot, w.bounds, w.pool)
        o, d = helpers.mixed_rays(w.dims, 1500, seed=10 + wi)
        c = w.trace_batch(o, d)
        for i in range(len(o)):
            r = ref_py.raytrace(pw, o[i], d[i])
            seen.update(r["events"])
            assert bool(c["hit"][i]) == r["hit"] and int(c["steps"][i]) == r["steps"], (wi, i)
            if r["hit"]:
                assert np.array_equal(np.array(r["pos"], f32).view(np.uint32), c["pos"][i].view(np.uint32)), (wi, i)
                assert np.array_equal(np.array(r["normal"], f32), c["normal"][i]), (wi, i)
                assert r["voxel"][0] + 64 * (r["voxel"][1] + 64 * r["voxel"][2]) == int(c["voxel"][i]), (wi, i)
    want = {"c:edge_pad", "b:edge_pad", "c:clamped_lookup", "b:clamped_lookup", "c:tie", "b:tie", "b:region_oob",
            "previous_cell_break", "ulp_nudge", "snap_x", "snap_y", "snap_z", "world_entry", "zero_steps", "normal_from_coarse"}
    assert want <= seen, want - seen
