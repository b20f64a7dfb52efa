from .actor_critic_decoder import ActorCriticDecoder, AC_Args
