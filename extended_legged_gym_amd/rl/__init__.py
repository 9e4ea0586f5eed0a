"""Rollout-collection side of PPO on the GPU: `NativeActorCritic`, `compute_returns`,
`collect_rollout`."""
from .policy import NativeActorCritic, NativeMLP          # noqa: F401
from .storage import compute_returns                      # noqa: F401
from .collector import collect_rollout                    # noqa: F401
