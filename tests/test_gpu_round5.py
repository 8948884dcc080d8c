"""Round-5 GPU tests: the guarded strips leg of bench.py under an ASYMMETRIC failure, the line contract of the mixed fp8 record."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_keeps_its_line_when_ONE_rank_fails_inside_the_strips_leg():
    """A rank that raises inside its seam exchange leaves its peer in the send / the mask all-gather; before round 5 the next collective
    then mismatched until the process-group timeout and the job died without a line.  Now the failing rank posts the reason in the
    store, every rank's watcher sees it, rank 0 prints the slides-leg line with strips_error and all ranks leave at once (two gloo
    ranks sharing the GPU; BENCH_STRIPS_FAIL_RANK=1 provokes the failure on rank 1 only)."""
    from conftest import cli_env
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1024", "--steps", "1", "--warmup", "1",
                          "--no-secondary"], capture_output=True, text=True, timeout=900,
                         env=cli_env(BENCH_DEVICE=0, BENCH_BACKEND="gloo", BENCH_STRIPS_FAIL_RANK=1, BENCH_STRIPS_DEADLINE_S=120))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "rank 1" in line["strips_error"] and "provoked failure" in line["strips_error"]
    assert len(out.stdout.strip().splitlines()) == 1                  # ONE JSON line
    assert time.time() - t0 < 300                                     # nowhere near the 240 s collective timeout after the failure


def test_bench_strips_leg_deadline_releases_a_hung_leg():
    """The other asymmetric case: nothing raises, the leg just does not finish (rank 1 sleeps inside its exchange).  The watcher's
    deadline prints the slides line."""
    from conftest import cli_env
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1024", "--steps", "1", "--warmup", "1",
                          "--no-secondary"], capture_output=True, text=True, timeout=900,
                         env=cli_env(BENCH_DEVICE=0, BENCH_BACKEND="gloo", BENCH_STRIPS_HANG_RANK=1, BENCH_STRIPS_DEADLINE_S=20))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["scaling"] == "weak" and "did not finish within" in line["strips_error"]


# ------------------------------------------------------------------------------------------- one image split by the window LIST
@pytest.mark.parametrize("shape,overlap,world", [((300, 520), 0.5, 2), ((700, 300), 0.5, 3), ((1000, 520), 0.25, 4), ((1024, 512), 0.5, 8),
                                                 ((200, 300), 0.5, 2), ((520, 520), 0.75, 5)])
def test_window_list_shares_are_bitwise_the_single_gpu_result(shape, overlap, world):
    """Every rank's share of the window LIST (cs_swi_partition_windows: shares differ by at most one window; the earlier windows it needs
    are evaluated locally in this single process) finalises its rectangles to exactly the bits of the single-GPU call - logits,
    probability and mask - and the rectangles tile the image."""
    import cellseg_amd as ca
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = ca.UNet(channels=(16, 32, 64), strides=(2, 2), num_res_units=2).to(dev).eval()
    img = torch.rand(3, *shape, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        logits, prob, mask = ca.swi_infer_image(model, img, (256, 256), overlap, want_logits=True, want_prob=True, want_mask=True)
        seen = torch.zeros(shape, dtype=torch.int32, device=dev)
        for r in range(world):
            sh, lg, pr, mk = ca.swi_infer_share(model, img, (256, 256), overlap, r, world, engine_batch=3, want_logits=True, want_prob=True,
                                                want_mask=True)
            for r0, r1, c0, c1 in sh.rects():
                seen[r0:r1, c0:c1] += 1
                assert torch.equal(lg[:, r0:r1, c0:c1], logits[:, r0:r1, c0:c1]), (r, r0, r1, c0, c1)
                assert torch.equal(pr[r0:r1, c0:c1], prob[r0:r1, c0:c1]) and torch.equal(mk[r0:r1, c0:c1], mask[r0:r1, c0:c1])
    assert bool((seen == 1).all())
