"""Child entry used by tests/test_bench_launch.py (never by the product): runs bench.main() -- the same argument parsing,
rank / shard arithmetic, barrier + max-over-ranks timing and JSON line -- on CPU ranks over gloo (bench.py's own default control
plane), with the CPU oracle (or, for the 8-rank rehearsal at the real job shapes, a null sampler) standing
in for the HIP sampler behind the begin() / step() / sample() interface bench.py drives.  The line it prints is marked
"backend: injected sampler"; it measures nothing about the product."""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
import oracle  # noqa: E402


class OracleStepSampler:
    def __init__(self, cfg, run, params, device):
        self.cfg, self.run, self.params = cfg, run, params
        self.use_graph = False

    def begin(self, cond, steps, *, seed=0, first_image=0, x_T=None):
        H = self.cfg["image_size"]
        x = torch.stack([oracle.sampler_ref.noise_ref(seed, first_image + b, 1, 1, (3, H, H))[0] for b in range(cond.shape[0])])
        return SimpleNamespace(plan=SimpleNamespace(xin=x), cond=cond, coefs=oracle.step_coefficients_ref(steps, self.run["method"]))

    def step(self, st, k):
        t, ca, cb, cx, c0, ce, sigma = st.coefs[k]
        x = st.plan.xin
        eps = oracle.unet_forward_ref(self.cfg, self.params, x, torch.full((x.shape[0],), t, dtype=torch.int64), st.cond)
        st.plan.xin = cx * x + c0 * (ca * x + cb * eps).clamp(-1, 1) + ce * eps      # (DDIM: sigma = 0)

    def sample(self, cond, steps, *, seed=0, first_image=0):
        return torch.stack([oracle.sample_ref(self.cfg, self.params, cond[k:k + 1], steps, seed=seed, method=self.run["method"],
                                              first_image=first_image + k)[0] for k in range(cond.shape[0])])


class NullStepSampler:
    """CDX_BENCH_STANDIN=null: same interface, no arithmetic worth the name (x <- 0.5 x per step, zeros for a decode): lets eight
    CPU ranks walk bench.main's launch / bind / shard / timing / JSON path at the REAL job shapes of cfg3 and cfg5 (16 images of
    256^2 / 8 images of 1024^2 as 25 tiles per rank), which the oracle could not decode in a test."""
    needs_params = False
    method = "ddim"

    def __init__(self, cfg, run, params, device):
        self.cfg, self.run, self.use_graph = cfg, run, False

    def tile_batch(self, cond, overlap, seed, first_image):
        import cdx
        T, H = self.cfg["image_size"], cond.shape[-1] * 16
        ys = cdx.tile_origins(H, T, overlap)
        n = cond.shape[0] * len(ys) ** 2
        return torch.zeros(n, cond.shape[1], T // 16, T // 16), torch.ones(n, 3, T, T), ys, ys

    def begin(self, cond, steps, *, seed=0, first_image=0, x_T=None):
        H = self.cfg["image_size"]
        x = torch.ones(cond.shape[0], 3, H, H) if x_T is None else x_T.clone()
        return SimpleNamespace(plan=SimpleNamespace(xin=x))

    def step(self, st, k):
        st.plan.xin = st.plan.xin * 0.5

    def sample(self, cond, steps, *, seed=0, first_image=0):
        H = self.cfg["image_size"]
        return torch.zeros(cond.shape[0], 3, H, H)

    def sample_tiled(self, cond, steps, *, overlap, seed=0, first_image=0):
        return torch.zeros(cond.shape[0], 3, cond.shape[-2] * 16, cond.shape[-1] * 16)


def mock_cuda(log_dir):
    """CDX_BENCH_BIND_LOG=dir: torch.cuda pretends a GPU per rank is there and records which device this rank binds."""
    def set_device(d):
        with open(os.path.join(log_dir, "rank%s.txt" % os.environ.get("RANK", "0")), "w") as f:
            f.write("%s %s" % (os.environ.get("LOCAL_RANK"), d))
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = set_device


if __name__ == "__main__":
    torch.set_num_threads(2 if int(os.environ.get("WORLD_SIZE", "1")) <= 2 else 1)
    log = os.environ.get("CDX_BENCH_BIND_LOG")
    if log:
        mock_cuda(log)
    standin = NullStepSampler if os.environ.get("CDX_BENCH_STANDIN") == "null" else OracleStepSampler
    bench.main(sys.argv[1:], make_sampler=standin, bind_device=bool(log))      # control plane: bench.py's default (gloo)
