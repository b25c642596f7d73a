"""BASELINE.json's full-size configurations, at the size that selects their kernels:

  configs[2]  2^16 mixed PPE + MSME equations, BLS12-381 (50 % PPE, 25 % MSMEG1, 25 % MSMEG2 -- SURVEY.md 8d)
  configs[3]  the 2^15-equation shard one GPU of eight owns of the 2^18 batch
  configs[4]  2^16 PPE, BN254

Each batch is proved with the planner left alone (so the large-batch shapes run: twin Miller lanes, 8-term Straus
groups, one lane per final exponentiation), >= 16 sampled equations -- first / last lanes of the first / last waves and
a spread in between -- are compared bit for bit with the C oracle, and the whole batch goes through the
size-independent properties (all honest proofs accepted, exactly the corrupted ones rejected, batched verdict agrees).
Reference: src/prover/prove.rs:92-171, src/verifier.rs:23-55."""
import pytest

from gpubatch import run_batch

pytestmark = pytest.mark.gpu


def spread(N, k=16):
    s = {0, 1, 63, 64, N // 2 - 1, N // 2, N - 65, N - 64, N - 2, N - 1}
    step = max(N // (k - len(s) + 1), 1)
    s.update(range(step // 2, N, step))
    return sorted(s)


# (k_var_multi*: the planner picks the group size together with the window width and the outputs per table build)
# (k_miller.pair*: two lanes per (equation, task) with one accumulator each -- what the planner picks at these sizes)
LARGE = ["k_miller.pair*", "k_final", "k_var_multi*.g1", "k_var_multi*.g2"]


def test_config2_mixed_2p16_bls12_381():
    N = 1 << 16
    run_batch(0, "bls12_381", 0, N // 2, 4, 4, spread(N // 2), seed=20241222, corrupt_every=1024, expect=LARGE)
    run_batch(0, "bls12_381", 1, N // 4, 4, 4, spread(N // 4), seed=20241222, corrupt_every=1024)
    run_batch(0, "bls12_381", 2, N // 4, 4, 4, spread(N // 4), seed=20241222, corrupt_every=1024)


def test_config3_shard_2p15_bls12_381():
    N = 1 << 15
    run_batch(0, "bls12_381", 0, N, 4, 4, spread(N), seed=20241223, corrupt_every=1024, expect=LARGE)


def test_config4_bn254_2p16():
    N = 1 << 16
    run_batch(1, "bn254", 0, N, 4, 4, spread(N), seed=20241224, corrupt_every=1024, expect=LARGE)


def test_config1_2p12_sampled_against_oracle():
    """configs[1] at its own size (the planner's mid-size shapes), 24 equations against the oracle."""
    N = 1 << 12
    run_batch(0, "bls12_381", 0, N, 4, 4, spread(N, 24), seed=20241221, corrupt_every=1024)
