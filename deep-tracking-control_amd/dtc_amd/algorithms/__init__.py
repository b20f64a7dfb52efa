from .ppo import PPO
from .recurrent_ppo import RecurrentPPO
from .recurrent_decoder_ppo import RecurrentDecoderPPO
