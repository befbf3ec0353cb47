"""Batch-invariant row placement for the library GEMMs of a sharded run (nn_ops.route_rows / route_batch): host logic,
no GPU.  The reference's DDP split of the camera batch (threestudio launch.py, GaussianDreamer.py:189-191) gives rank r
the views r, r + k, ...; the single-rank batch of the SDS UNet call is [text views | unconditional views]."""
import pytest
import torch

from garmentdreamer_amd import nn_ops


@pytest.fixture
def scale2():
    yield
    nn_ops.set_route_scale(1)


def test_rows_land_where_the_single_rank_batch_holds_them(scale2):
    k, c, T, K = 2, 3, 5, 4                       # 2 ranks, 3 views per rank, 5 tokens per sample
    whole = torch.arange(2 * k * c * T * K, dtype=torch.float32).view(2 * k * c * T, K)     # [text 6 views | uncond 6 views]
    W = torch.randn(7, K)
    for r in range(k):
        nn_ops.set_route_scale(k, r)
        views = list(range(r, k * c, k))
        mine = torch.cat([whole.view(2, k * c, T, K)[g, views] for g in range(2)]).reshape(-1, K)
        with nn_ops.route_batch(2, 2 * c):
            padded, take = nn_ops.route_rows(mine)
        assert padded.shape == whole.shape
        keep = torch.zeros(2, k * c, dtype=torch.bool)
        keep[:, views] = True
        keep = keep[:, :, None].expand(2, k * c, T).reshape(-1)
        assert torch.equal(padded[keep], whole[keep]) and not padded[~keep].any()
        assert torch.equal(take(padded @ W.t()), mine @ W.t())


def test_unknown_batch_structure_pads_behind_the_rows(scale2):
    nn_ops.set_route_scale(3, 1)
    rows = torch.randn(10, 4)
    padded, take = nn_ops.route_rows(rows)          # no route_batch: this rank's rows at share 1 of 3, zeros elsewhere
    assert padded.shape == (30, 4) and torch.equal(take(padded), rows)
    with nn_ops.route_batch(2, 4):                   # 10 rows do not divide into 4 samples: same fallback
        p2, t2 = nn_ops.route_rows(rows)
    assert torch.equal(p2, padded) and torch.equal(t2(p2), rows)


def test_route_scale_arguments_are_checked(scale2):
    with pytest.raises(ValueError):
        nn_ops.set_route_scale(2, 2)
    with pytest.raises(ValueError):
        nn_ops.set_route_scale(0)
    assert nn_ops.route_scale() == 1
