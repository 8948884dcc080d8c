"""Pin the CPU oracle against fixtures produced by real scikit-image 0.18.3 and by the imported
reference functions (generators: tests/golden/make_*.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import f1_ref, label_contract_ref, normalize_ref
from oracle import postprocess_ref as P

G = os.path.join(os.path.dirname(__file__), "golden")


def _cases(npz):
    return sorted({k.split("/")[0] for k in npz.files})


def _unpack(g, n, key):
    h, w = g[n + "/shape"]
    return np.unpackbits(g[n + "/" + key])[: h * w].reshape(h, w).astype(bool)


POST = np.load(os.path.join(G, "postprocess_golden.npz"))


@pytest.mark.parametrize("name", _cases(POST))
@pytest.mark.parametrize("impl", ["scipy", "c"])
def test_postprocess_matches_skimage(name, impl):
    m = _unpack(POST, name, "mask")
    lab = (lambda m, c: P.label_scipy(m, c)) if impl == "scipy" else (lambda m, c: P.label_c(m, c))
    post = P.mask_postprocess_scipy if impl == "scipy" else P.mask_postprocess_c
    assert np.array_equal(lab(m, 1), POST[name + "/label4_raw"])
    assert np.array_equal(lab(m, 2), POST[name + "/label8_raw"])
    assert np.array_equal(post(m), POST[name + "/labels"])       # bit-exact incl. raster-order ids
    assert np.array_equal(P.remove_small_holes_ref(m), _unpack(POST, name, "holes"))
    assert np.array_equal(P.remove_small_objects_ref(_unpack(POST, name, "holes"), 16), _unpack(POST, name, "objs"))


def test_threshold_is_strict():
    prob = np.array([[0.5, 0.50001, 0.49999]], np.float32).repeat(20, 0)
    assert (P.postprocess_scipy(prob, hole_area=0, min_size=0) > 0).sum() == 20


F1 = np.load(os.path.join(G, "f1_golden.npz"))


@pytest.mark.parametrize("name", _cases(F1))
def test_f1_matches_reference(name):
    gt, seg = F1[name + "/gt"], F1[name + "/seg"]
    assert tuple(f1_ref.eval_tp_fp_fn_ref(gt, seg)) == tuple(F1[name + "/tpfpfn"])
    gt_rb, seg_rb = f1_ref.remove_boundary_cells_ref(gt), f1_ref.remove_boundary_cells_ref(seg)
    assert np.array_equal(gt_rb, F1[name + "/gt_rb"]) and np.array_equal(seg_rb, F1[name + "/seg_rb"])
    assert tuple(f1_ref.eval_tp_fp_fn_ref(gt_rb, seg_rb)) == tuple(F1[name + "/tpfpfn_rb"])
    assert f1_ref.dice_ref(gt, seg) == pytest.approx(float(F1[name + "/dice"]), abs=0)


F1P = np.load(os.path.join(G, "f1_patch_golden.npz"))


@pytest.mark.parametrize("name", _cases(F1P))
def test_f1_patch_walk_matches_reference(name):
    got = f1_ref.f1_patch_counts_ref(F1P[name + "/gt"], F1P[name + "/seg"], int(F1P[name + "/edge"]))
    assert tuple(got) == tuple(F1P[name + "/totals"])


NORM = np.load(os.path.join(G, "normalize_golden.npz"))


@pytest.mark.parametrize("name", _cases(NORM))
def test_normalize_matches_reference(name):
    out = normalize_ref.normalize_channel_ref(NORM[name + "/in"])
    assert out.dtype == np.uint8 and np.array_equal(out, NORM[name + "/out"])


LC = np.load(os.path.join(G, "label_contract_golden.npz"))


@pytest.mark.parametrize("name", _cases(LC))
def test_label_contract_matches_reference(name):
    assert np.array_equal(label_contract_ref.create_interior_map_ref(LC[name + "/inst"]), LC[name + "/interior"])


# ----------------------------------------------------------------------------------------------------------------------
# the reference's own three sample pairs (data/Train_Labeled): outputs of the IMPORTED reference functions on real data
# (tests/golden/make_sample_golden.py); SURVEY 8(c) G6 measured cell_00849 -> 143/13/8 and cell_00225 -> 5/0/0 with the boundary rule
# ----------------------------------------------------------------------------------------------------------------------
SAMPLES = np.load(os.path.join(G, "sample_golden.npz"))


def test_sample_goldens_hold_the_surveyed_counts():
    assert tuple(SAMPLES["cell_00849/tpfpfn_rb"]) == (143, 13, 8) and tuple(SAMPLES["cell_00225/tpfpfn_rb"]) == (5, 0, 0)
    assert SAMPLES["cell_00528/img"].dtype == np.uint16 and SAMPLES["cell_00528/img"].shape == (200, 200)      # the pad-to-256 input
    assert int(SAMPLES["cell_00849/gt"].max()) == 187


@pytest.mark.parametrize("name", _cases(SAMPLES))
def test_oracle_matches_reference_on_its_own_samples(name):
    img, gt = SAMPLES[name + "/img"], SAMPLES[name + "/gt"]
    pre = normalize_ref.preprocess_image_ref(img)                                    # predict.py:94-104 == pre_process_3class.py:84-95
    assert pre.dtype == np.uint8 and np.array_equal(pre, SAMPLES[name + "/pre"])
    interior = label_contract_ref.create_interior_map_ref(gt)                        # pre_process_3class.py:31-56
    assert np.array_equal(interior, SAMPLES[name + "/interior"])
    seg = SAMPLES[name + "/seg"]
    assert np.array_equal(P.label_scipy(interior == 1, 2), seg) and np.array_equal(P.label_c(interior == 1, 2), seg)   # measure.label
    assert tuple(f1_ref.eval_tp_fp_fn_ref(gt, seg)) == tuple(SAMPLES[name + "/tpfpfn"])                                 # compute_metric.py
    gt_rb, seg_rb = f1_ref.remove_boundary_cells_ref(gt), f1_ref.remove_boundary_cells_ref(seg)
    assert tuple(f1_ref.eval_tp_fp_fn_ref(gt_rb, seg_rb)) == tuple(SAMPLES[name + "/tpfpfn_rb"])
    for post in (P.mask_postprocess_scipy, P.mask_postprocess_c):                    # predict.py:113 on a real interior mask
        assert np.array_equal(post(interior == 1), SAMPLES[name + "/label_m"])
