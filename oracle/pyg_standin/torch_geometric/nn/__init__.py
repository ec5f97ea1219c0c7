from .message_passing import MessagePassing
from .glob import global_add_pool, global_mean_pool, global_max_pool
from . import inits, glob  # noqa: F401
