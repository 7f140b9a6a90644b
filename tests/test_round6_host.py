"""Host logic of round 6 that needs no GPU: the chunking of the module API's autograd chain and which chunk owns which parameter
gradient (hsenet_amd/fused.py), the witnesses that tell a 16-bit weight copy is stale (functional.weights_stamp), and bench.py's
device count without a GPU context."""
import os

import torch

from hsenet_amd import fused
from hsenet_amd import functional as Fn


def test_chunks_partition_the_layers_and_every_gradient_has_one_owner():
    for n_layers, n_chunks in ((12, 4), (12, 1), (2, 4), (6, 4), (5, 2), (1, 4), (7, 3)):
        r = fused.chunk_ranges(n_layers, n_chunks)
        assert r[0][0] == 0 and r[-1][1] == n_layers and all(a[1] == b[0] for a, b in zip(r, r[1:])) and len(r) <= max(n_chunks, 1) + 0
        slots = [s for lo, hi in r for s in fused.owned_slots(n_layers, lo, hi)]
        # 11 tensors per block + the final norm + the 16 embedding-driver slots, each exactly once
        assert len(slots) == len(set(slots)) == n_layers * 11 + 2 + 16
        for lo, hi in r:
            own = fused.owned_slots(n_layers, lo, hi)
            # linear2.bias (index 3) of a chunk's TOP layer is finished by the chunk above -- except at the very top, where the final norm does it
            assert (("b", hi - 1, 3) in own) == (hi == n_layers)
            assert (("b", lo - 1, 3) in own) == (lo > 0)
            assert (("n", 0, 0) in own) == (hi == n_layers) and (("h", 0, 0) in own) == (lo == 0)
    # the engines' bucket split is the same split seen from the other end
    from hsenet_amd.engine import bucket_plan
    names = [f"blocks.{l}.mlp.linear1.weight" for l in range(12)]
    chunks, _ = bucket_plan(names, list(range(0, 1200, 100)), 1300, 12, 4)
    assert [(lo, hi) for hi, lo in chunks][::-1] == fused.chunk_ranges(12, 4)


def test_weight_copy_staleness_witnesses():
    """version counters see in-place writes through the parameter or a detach()ed alias; torch's FUSED optimizer kernels do not touch them,
    the global optimizer post-step hook does; `.data` writes need invalidate_weight_copies()."""
    import hsenet_amd as H
    ps = [torch.nn.Parameter(torch.randn(4, 4)) for _ in range(3)]
    frozen = [torch.nn.Parameter(torch.randn(4, 4), requires_grad=False)]
    s0, f0 = Fn.weights_stamp(ps), Fn.weights_stamp(frozen)
    assert Fn.weights_stamp(ps) == s0
    for fused_opt in (True, False):
        opt = torch.optim.AdamW(ps, lr=1e-2, fused=fused_opt)
        sum((q ** 2).sum() for q in ps).backward()
        before = Fn.weights_stamp(ps)
        opt.step()
        assert Fn.weights_stamp(ps) != before
        opt.zero_grad(set_to_none=True)
    assert Fn.weights_stamp(frozen) == f0                       # an optimizer that does not own them leaves other modules' copies alone
    s1 = Fn.weights_stamp(ps)
    with torch.no_grad():
        ps[1].mul_(2.0)
    s2 = Fn.weights_stamp(ps)
    assert s2 != s1
    lin = torch.nn.Linear(4, 4)
    l0 = Fn.weights_stamp([lin.weight])
    lin.state_dict()["weight"].copy_(torch.ones(4, 4))          # the reference's positional checkpoint hand-over (train_VLM.py:477-503)
    assert Fn.weights_stamp([lin.weight]) != l0
    l1 = Fn.weights_stamp([lin.weight])
    lin.load_state_dict({"weight": torch.zeros(4, 4), "bias": torch.zeros(4)})
    assert Fn.weights_stamp([lin.weight]) != l1
    ps[2].data.mul_(2.0)                                        # an alias with its own version counter: invisible ...
    assert Fn.weights_stamp(ps) == s2
    H.invalidate_weight_copies()                                # ... hence the explicit call
    assert Fn.weights_stamp(ps) != s2 and Fn.weights_stamp(frozen) != f0


def test_visible_gpus_counts_without_touching_the_runtime(monkeypatch, tmp_path):
    import bench
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() <= max(1, n)
    assert os.environ.get("HS_KEEP_GRADS") == "1"               # (tests/conftest.py: the suite inspects gradients after optimizer steps)
