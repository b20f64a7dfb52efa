from .actor_critic_decoder import ActorCriticDecoder, AC_Args
from .actor_critic_recurrent import ActorCritic, ActorCriticRecurrent, Memory
