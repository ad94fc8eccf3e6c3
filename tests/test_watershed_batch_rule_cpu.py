"""The batched flood's rule (several pops per round, tests/_ws_batch_model.py = what csrc/ct_segment.hip::ws_flood_batch_kernel does) gives the
sequential priority flood's labels -- checked on the CPU against the oracle's restatement of skimage.segmentation.watershed."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import watershed_ref as wr
from _ws_batch_model import batched_flood
from _ws_cases import touching_case, random_case, wide_front_case, tie_case


def _stage_inputs(prob, three_d):
    """(image, markers, mask) of the reference's watershed calls: every slice of watershed_2d, or watershed_3d's single call"""
    if not three_d:
        for z in range(prob.shape[2]):
            bn = prob[:, :, z] > 0.5
            sm = ndi.gaussian_filter(ndi.distance_transform_edt(bn, sampling=[1, 1]), 2, mode="constant")
            yield -sm, wr.label_full(wr.peak_local_max_mask(sm, min_distance=7)), bn
    else:
        bn, _ = wr.watershed_2d(prob, z_range=prob.shape[2], min_distance=7)
        sm = ndi.gaussian_filter(ndi.distance_transform_edt(bn, sampling=[1, 1, 3.0]), (2, 2, 0.3), mode="constant")
        yield -sm, wr.label_full(wr.peak_local_max_mask(sm, min_distance=3, exclude_border=0)), bn


CASES = {"touching": touching_case, "random": lambda: random_case((60, 50, 8), 14, 5), "wide_front": wide_front_case, "ties": tie_case}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("three_d", [False, True], ids=["2d", "3d"])
def test_batched_rule_equals_the_sequential_flood(name, three_d):
    prob = CASES[name]()
    total_rounds = total_pops = 0
    for image, markers, mask in _stage_inputs(prob, three_d):
        want = wr.watershed(image, markers, mask, seed_order="raveled")       # (the device's rule among exactly equal seeds; upstream's order
        for lanes, cap in (((64, 256),) if name == "wide_front" and three_d else ((64, 256), (4, 3))):   #  is a separate replay, tests/test_watershed_pin.py)
            got, rounds, pops = batched_flood(image, markers, mask, lanes=lanes, member_cap=cap)
            assert np.array_equal(got, want), f"{name}: {int((got != want).sum())} voxels differ (lanes {lanes}, cap {cap})"
            if lanes == 64:
                total_rounds += rounds; total_pops += pops
    assert total_pops >= total_rounds


def test_batched_rule_needs_far_fewer_rounds_than_pops_on_a_wide_front():
    prob = wide_front_case()
    image, markers, mask = next(_stage_inputs(prob, False))
    _, rounds, pops = batched_flood(image, markers, mask)
    assert pops > 4000 and rounds * 8 < pops, (rounds, pops)
