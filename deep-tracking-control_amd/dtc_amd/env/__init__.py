from .vec_env import VecEnv, ReplayEnv
from .wrappers.history_wrapper import HistoryWrapper
