from . import conv  # noqa: F401
from .conv import DotGatConv, SAGEConv, GATConv, GATv2Conv  # noqa: F401
