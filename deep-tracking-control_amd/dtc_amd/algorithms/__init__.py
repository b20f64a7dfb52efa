from .ppo import PPO
from .recurrent_ppo import RecurrentPPO
