from .message_passing import MessagePassing  # noqa: F401
