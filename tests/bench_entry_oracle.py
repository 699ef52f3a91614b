"""Child entry used by tests/test_bench_launch.py (never by the product): runs bench.main() -- the same argument parsing,
rank / shard arithmetic, barrier + max-over-ranks timing and JSON line -- on CPU ranks over gloo, with the CPU oracle standing
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


if __name__ == "__main__":
    torch.set_num_threads(2)
    bench.main(sys.argv[1:], make_sampler=OracleStepSampler, dist_backend="gloo")
