"""CPU: host-side logic of the frame runners that needs no GPU."""
import numpy as np
import torch


def test_refinement_time_step_follows_the_bank_rule():
    """runner.refinement_time_step (what SplitPipelinedRunner stages for the single-frame decoder layer, which runs before
    InstanceBank.get of its frame) against the oracle's statement of instance_bank.py:87,108-113: dt where it is non-zero
    and within max_time_interval, the default time step otherwise; f32 throughout."""
    from simpb_amd.runner import refinement_time_step
    max_dt, default = 2.0, 0.5
    dt = np.array([0.5, 0.0, 2.0, 2.0000002, -0.5, -2.5, 1e-30, 7.25, np.float32(1.9999999)], np.float32)
    got = refinement_time_step(dt, max_dt, default)
    t = torch.from_numpy(dt)
    mask = t.abs() <= max_dt                                                        # oracle/simpb_ref.py InstanceBank.get
    want = torch.where((t != 0) & mask, t, t.new_tensor(default)).numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert got[1] == np.float32(default) and got[3] == np.float32(default) and got[4] == np.float32(-0.5)
