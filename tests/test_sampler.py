"""Train-split ray sampler.  CPU: the oracle restatement against the fixture recorded from the REAL
PhototourismDataset.__getitem__ (tools/make_goldens.py:sampler_fixture).  GPU: upnerf_gather_rays through
upnerf_amd.ray_sampler.GpuRaySampler against that fixture and against the oracle, bit for bit."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, orc

KEYS = ("ray_infos", "directions", "img_idx", "c2w", "rgbs", "feats", "inv_depths")
BUFS = ("all_ray_infos", "all_directions", "all_rgbs", "all_pxl_coords", "all_inv_depths", "feat_maps", "poses")


def _fixture():
    g = np.load(os.path.join(GOLDEN, "sampler.npz"))
    return {k: torch.from_numpy(g[k]) for k in g.files}


def test_oracle_sampler_matches_reference_getitem():
    g = _fixture()
    out = orc.sample_train_rays({k: g[k] for k in BUFS}, g["idx"])
    for k in KEYS:
        assert out[k].dtype == g["out_" + k].dtype and torch.equal(out[k], g["out_" + k]), k
    # the reference's interpolation weights vanish on the last row / column of a feature map: those rays get zeros
    last = (g["all_pxl_coords"][g["idx"]] == 1.0).any(1)
    assert bool(last.any()) and float(out["feats"][last].abs().max()) == 0.0
    assert float(out["feats"][~last].abs().sum(1).min()) > 0.0


def _sampler(bufs):
    from upnerf_amd.ray_sampler import GpuRaySampler
    return GpuRaySampler(bufs["all_ray_infos"], bufs["all_directions"], bufs["all_rgbs"], bufs["poses"],
                         all_pxl_coords=bufs["all_pxl_coords"], feat_maps=bufs["feat_maps"],
                         all_inv_depths=bufs["all_inv_depths"])


@pytest.mark.gpu
def test_gpu_sampler_is_bit_exact_with_the_reference_fixture():
    g = _fixture()
    out = _sampler(g).sample(g["idx"].cuda())
    for k in KEYS:
        assert torch.equal(out[k].cpu(), g["out_" + k]), k


@pytest.mark.gpu
def test_gpu_sampler_matches_oracle_on_a_larger_random_set_and_shards_like_a_distributed_sampler():
    gen = torch.Generator().manual_seed(3)
    I, h, C, N = 5, 9, 384, 3000
    bufs = {"all_ray_infos": torch.cat([torch.rand(N, 2, generator=gen), torch.randint(0, I, (N, 1), generator=gen).float()], 1),
            "all_directions": torch.randn(N, 3, generator=gen), "all_rgbs": torch.rand(N, 3, generator=gen),
            "all_pxl_coords": torch.rand(N, 2, generator=gen), "all_inv_depths": torch.rand(N, generator=gen),
            "feat_maps": torch.randn(I, h, h, C, generator=gen), "poses": torch.randn(I, 3, 4, generator=gen)}
    bufs["all_pxl_coords"][:50] = torch.randint(0, 2, (50, 2), generator=gen).float()  # exact corners and edges
    smp = _sampler(bufs)
    idx = torch.randint(0, N, (777,), generator=gen)
    ref = orc.sample_train_rays(bufs, idx)
    out = smp.sample(idx.cuda())
    for k in KEYS:
        assert torch.equal(out[k].cpu(), ref[k]), k
    # one epoch over two ranks: disjoint, complete, per-rank batch size, same permutation on both ranks
    seen = []
    for rank in (0, 1):
        rows = [b["directions"].cpu() for b in smp.batches(256, seed=7, epoch=1, rank=rank, world_size=2)]
        assert all(r.shape[0] <= 256 for r in rows)
        seen.append(torch.cat(rows))
    allrows = torch.cat(seen)
    assert seen[0].shape[0] == seen[1].shape[0] == -(-N // 2) and allrows.shape[0] == N  # N is even: no padding
    assert torch.equal(torch.sort(allrows[:, 0])[0], torch.sort(bufs["all_directions"][:, 0])[0])


@pytest.mark.gpu
def test_gpu_sampler_feeds_the_training_step():
    """A sampled batch has the keys / dtypes NeRFSystem.training_step consumes (phototourism.py:421-454 -> nerf_system.py:150-166)."""
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    gen = torch.Generator().manual_seed(5)
    I, h, N = 4, 8, 2000
    d = torch.randn(N, 3, generator=gen)
    d[:, 2] = -1.0
    bufs = {"all_ray_infos": torch.cat([torch.full((N, 1), 0.1), torch.full((N, 1), 5.0), torch.randint(0, I, (N, 1), generator=gen).float()], 1),
            "all_directions": d, "all_rgbs": torch.rand(N, 3, generator=gen), "all_pxl_coords": torch.rand(N, 2, generator=gen),
            "all_inv_depths": torch.rand(N, generator=gen) * 2 + 0.3,
            "feat_maps": torch.nn.functional.normalize(torch.randn(I, h, h, 384, generator=gen), dim=-1),
            "poses": torch.eye(3, 4).repeat(I, 1, 1)}
    smp = _sampler(bufs)
    hp = default_hparams(**{"nerf.N_samples": 32, "nerf.N_importance": 32, "train.batch_size": 128})
    torch.manual_seed(0)
    sysm = NeRFSystem(hp, SyntheticDataset(I))
    sysm.setup()
    sysm.cuda()
    sysm.set_progress(0.3)
    batch = next(iter(smp.batches(128, seed=1)))
    loss = sysm.training_step(batch, 0)
    assert torch.isfinite(loss)


def test_every_rank_gets_the_same_number_of_batches():
    """DistributedSampler semantics (train.py:70-72 via Lightning): for every remainder position of the last global batch
    all ranks run the same number of batches, the union of their indices is the whole split, and the padding is the
    wrapped head of the permutation."""
    from upnerf_amd.ray_sampler import epoch_indices
    bs, world = 256, 8
    for N in (3000, 2048 * 3 + 5, 2048 * 3 + 255, 2048 * 3 + 257, 2048 * 3 + 1800, 7):
        per = [epoch_indices(N, seed=3, epoch=2, rank=r, world_size=world, device="cpu") for r in range(world)]
        assert len({p.numel() for p in per}) == 1 and per[0].numel() == -(-N // world)
        counts = {-(-p.numel() // bs) for p in per}
        assert len(counts) == 1
        allidx = torch.stack(per, 1).reshape(-1)  # interleaved = the padded permutation
        full = epoch_indices(N, seed=3, epoch=2, device="cpu")
        assert torch.equal(allidx[:N], full)
        assert torch.equal(allidx[N:], full[:allidx.numel() - N])
        assert torch.equal(torch.sort(torch.unique(allidx))[0], torch.arange(N))
