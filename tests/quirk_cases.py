"""Hand-derived known-answer rays for the quirks of the reference's traversal that SURVEY.md 8a lists as "semantics to
preserve" (VoxelRT/VolumeRaytracer.cu).  Each case was stepped through the reference's text by hand (the derivations are
in tests/test_oracle_quirk_kat.py's docstrings); `events` are the tracing labels of oracle/ref_py.py that prove the ray
reaches the lines it is named for.  Used on the CPU against both restatements of the oracle and on the GPU against the HIP
path (tests/test_gpu_parity.py::test_quirk_cases_on_gpu).

A case: name -> dict(size=(X,Y,Z) voxels, factor, voxels=[(x,y,z)...], origin, dir, max_steps,
                     expect=dict(hit, steps, pos, normal, voxel, stats=(coarse probes, brick entries, brick probes)), events)
"""
import numpy as np

f32 = np.float32
DENORM = float(np.nextafter(f32(0), f32(-1)))          # nextafterf(0, -inf)
UP22 = float(f32(2.0) ** f32(-22))                     # one ulp in [2, 4)
UP23 = float(f32(2.0) ** f32(-23))                     # one ulp in [1, 2)


def voxel_index(v, size):
    return v[0] + size[0] * (v[1] + size[1] * v[2])


CASES = {
    # --- maxSteps is only tested at the head of the two-level loop (:386); the bounce rays run with maxSteps = 8 (Renderer.cu:141)
    "soft_max_steps_overrun": dict(
        size=(64, 64, 64), factor=8, voxels=[(21, 12, 11), (21, 10, 11), (29, 11, 11)],
        origin=(4.0, 11.5, 11.5), dir=(1, 0, 0), max_steps=8,
        expect=dict(hit=True, steps=10, pos=(29.0, 11.5, 11.5), normal=(1, 0, 0), voxel=(29, 11, 11), stats=(4, 2, 9)),
        events=[]),
    "soft_max_steps_cut": dict(
        size=(64, 64, 64), factor=8, voxels=[(53, 12, 11), (53, 10, 11), (61, 11, 11)],
        origin=(0.5, 11.5, 11.5), dir=(1, 0, 0), max_steps=8,
        expect=dict(hit=False, steps=9, pos=None, normal=(0, 0, 0), voxel=None, stats=(7, 1, 3)),
        events=[]),
    "soft_max_steps_cut_unbounded": dict(      # the same ray with MAX_STEPS: it goes on and hits the next brick
        size=(64, 64, 64), factor=8, voxels=[(53, 12, 11), (53, 10, 11), (61, 11, 11)],
        origin=(0.5, 11.5, 11.5), dir=(1, 0, 0), max_steps=2048,
        expect=dict(hit=True, steps=14, pos=(61.0, 11.5, 11.5), normal=(1, 0, 0), voxel=(61, 11, 11), stats=(8, 2, 9)),
        events=["c:box_hit_at_step0"]),
    # --- world entry box [1e-6, C - 1e-6]^3 with the double-typed FLT_EPS_DDA rounded to float once (:373-376):
    #     C = 64 -> the far face is 64.0 itself and the ray starts in cell 64 == dim (edge rule, :216-232, coarse AND brick);
    #     C = 16 -> 15.99999905 and an ordinary start in cell 15
    "world_entry_dim64_edge_rule": dict(
        size=(512, 64, 64), factor=8, voxels=[(511, 11, 11)],
        origin=(600.0, 11.5, 11.5), dir=(-1, 0, 0), max_steps=2048,
        expect=dict(hit=True, steps=0, pos=(512.0, 11.5, 11.5), normal=(-1, 0, 0), voxel=(511, 11, 11), stats=(1, 1, 1)),
        events=["outside_start", "world_entry", "c:edge", "c:edge_pad", "c:clamped_lookup", "c:box_hit_at_step0", "b:edge",
                "b:edge_pad", "b:clamped_lookup", "normal_from_coarse", "zero_steps"]),
    "world_entry_dim16_no_edge": dict(
        size=(128, 64, 64), factor=8, voxels=[(127, 11, 11)],
        origin=(200.0, 11.5, 11.5), dir=(-1, 0, 0), max_steps=2048,
        expect=dict(hit=True, steps=0, pos=(float(np.nextafter(f32(128), f32(0))), 11.5, 11.5), normal=(-1, 0, 0),
                    voxel=(127, 11, 11), stats=(1, 1, 1)),
        events=["outside_start", "world_entry", "c:box_hit_at_step0", "normal_from_coarse", "zero_steps"]),
    # --- a ray sliding down the far face x = 64 (coarse x = 8.0 == dim): edge padding on the coarse grid and in the brick,
    #     the clamped cell probed twice, the -0 == -0 tie taken by y (:293-313), no nudge because int(8.0) != 7 (:445-447), and the
    #     restart hits the same coarse cell again -> previous_cell break, a hole (:402-407)
    "edge_slide_previous_cell_hole": dict(
        size=(64, 64, 64), factor=8, voxels=[(63, 13, 10), (63, 8, 12)],
        origin=(64.0, 20.5, 11.5), dir=(-1e-30, -1, 0), max_steps=2048,
        expect=dict(hit=False, steps=10, pos=None, normal=(0, 0, 0), voxel=None, stats=(4, 1, 8)),
        events=["outside_start", "c:edge", "c:edge_pad", "c:clamped_lookup", "b:edge", "b:edge_pad", "b:clamped_lookup", "b:tie",
                "c:tie", "c:box_hit_at_step0", "previous_cell_break"]),
    # --- leaving the world through a face at coordinate 0: the restart point truncates to the same coarse cell, the one-ulp
    #     nudge lands on -denormal which still truncates to 0, so one axis is snapped to NextCell (:470-487); which axis:
    #     x if strictly smallest, else y if strictly smallest, else z (ties go to z)
    "floor_exit_snap_z": dict(
        size=(64, 64, 64), factor=8, voxels=[(21, 3, 10), (21, 3, 12)],
        origin=(21.5, 20.5, 11.5), dir=(0, -1, 0), max_steps=2048,
        expect=dict(hit=False, steps=8, pos=None, normal=(0, 0, 0), voxel=None, stats=(4, 1, 5)),
        events=["ulp_nudge", "snap_z"]),
    "floor_exit_snap_x": dict(
        size=(64, 64, 64), factor=8, voxels=[(17, 3, 10), (17, 3, 12)],
        origin=(17.5, 20.5, 11.5), dir=(0, -1, 0), max_steps=2048,
        expect=dict(hit=False, steps=8, pos=None, normal=(0, 0, 0), voxel=None, stats=(4, 1, 5)),
        events=["ulp_nudge", "snap_x"]),
    "wall_exit_snap_y": dict(
        size=(64, 64, 64), factor=8, voxels=[(3, 10, 11), (3, 12, 11)],
        origin=(20.5, 11.5, 11.75), dir=(-1, 0, 0), max_steps=2048,
        expect=dict(hit=False, steps=8, pos=None, normal=(0, 0, 0), voxel=None, stats=(4, 1, 5)),
        events=["ulp_nudge", "snap_y"]),
    "wall_exit_snap_tie_goes_to_z": dict(
        size=(64, 64, 64), factor=8, voxels=[(3, 10, 11), (3, 12, 11)],
        origin=(20.5, 11.5, 11.5), dir=(-1, 0, 0), max_steps=2048,
        expect=dict(hit=False, steps=8, pos=None, normal=(0, 0, 0), voxel=None, stats=(4, 1, 5)),
        events=["ulp_nudge", "snap_z"]),
}

# Found by search over seeded rays (oracle/ref_py.py events), then frozen bit for bit; world = default_rng(0).random(64^3) < 0.02,
# factor 8.  The brick-level region check (:325-341) ending a walk from inside the world is a rounding-level event that
# cannot be built from round numbers: this ray runs along the exact x == y diagonal (every step a tie), one brick walk
# ends by the region check, the restart needs the ulp nudge AND the NextCell snap, and the ray ends on the world's far
# face under the edge rule.  Expected values: the common answer of the two restatements, frozen.
SEARCHED = {
    "diagonal_ties_region_oob_nudge_snap_edge": dict(
        world_seed=0, density=0.02, size=(64, 64, 64), factor=8,
        origin_bits=(0x42c00000, 0x42c00000, 0x42082761), dir_bits=(0xbf0a6397, 0xbf0a6397, 0x3eb37dc5), max_steps=2048,
        expect=dict(hit=False, steps=39, pos=None, normal=(0, 0, 0), voxel=None, stats=(5, 3, 38)),
        events=["outside_start", "world_entry", "c:tie", "b:tie", "b:region_oob", "ulp_nudge", "snap_x", "c:edge", "c:edge_pad"]),
}


def build_case(vxo, case):
    """(oracle world, origin float32[3], dir float32[3]) of a CASES / SEARCHED entry."""
    if "voxels" in case:
        v = np.zeros(case["size"], bool)
        for p in case["voxels"]:
            v[p] = True
        o, d = np.array(case["origin"], f32), np.array(case["dir"], f32)
    else:
        v = np.random.default_rng(case["world_seed"]).random(case["size"]) < case["density"]
        o = np.array(case["origin_bits"], np.uint32).view(f32)
        d = np.array(case["dir_bits"], np.uint32).view(f32)
    return vxo.World.from_voxels(v, case["factor"]), o, d


def all_cases():
    out = dict(CASES)
    out.update(SEARCHED)
    return out
