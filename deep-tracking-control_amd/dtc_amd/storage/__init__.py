from .rollout_storage import RolloutStorage
