import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
from upnerf_amd import synth
from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
dev = torch.device("cuda", 0)
for name, over, R in [("default.yaml shape (128+128, batch 2048)", {}, 2048),
                      ("ragged batch 1000, 64+128", {"nerf.N_samples": 64, "nerf.N_importance": 128}, 1000),
                      ("no fine pass 64+0", {"nerf.N_samples": 64, "nerf.N_importance": 0}, 4096),
                      ("use_disp, perturb 0", {"nerf.N_samples": 64, "nerf.N_importance": 64, "nerf.use_disp": True, "nerf.perturb": 0.0}, 512),
                      ("pose.optimize off", {"nerf.N_samples": 64, "nerf.N_importance": 128, "pose.optimize": False}, 2048)]:
    hp = default_hparams(**over)
    torch.manual_seed(0)
    sysm = NeRFSystem(hp, SyntheticDataset(763)); sysm.setup(); sysm.to(dev)
    for prog in (0.05, 0.3, 0.8):
        sysm.global_step = int(prog * 2 * hp["max_steps"]); sysm.set_progress(prog)
        b = {k: v.to(dev) for k, v in synth.batch(R, 763, seed=7).items()}
        for i in range(3):
            loss = sysm.training_step(b, i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5): loss = sysm.training_step(b, i)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(loss)) and all(bool(torch.isfinite(p).all()) for p in sysm.parameters())
        print(f"{name:45s} progress {prog}: loss {float(loss):9.5f} finite={ok} {(time.perf_counter() - t0) / 5 * 1e3:6.2f} ms/step  {R / ((time.perf_counter() - t0) / 5):9.0f} rays/s")
        assert ok
