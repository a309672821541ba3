"""gwen_amd -- MI355X-native implementation of GWEN's GCNConv-stack hot path.

Public surface mirrors /root/reference/src/gwen/models_gnn.py (GNNConfig, DownConvLayers,
UpConvLayers, GCNConvLayers, GNNModel, loss_func) plus the ``GCNConv`` layer it imports from
torch-geometric (:19).  Kernels live in libgwen_hip.so (include/gwen_hip.h); build it with
``python -m gwen_amd.build``.
"""
from . import forecaster, g2m, interaction, ops
from .forward import GraphedForward, KernelEvents, StackForward, event_bracket_overhead
from .gcn_conv import GCNConv, Linear
from .forecaster import InteractionForecaster
from .interaction import EdgeGraph, InteractionNet, interaction_graph
from .graph import GraphCSR, GraphCache, default_cache, prepare_graph
from .mesh import Mesh, complete_graph, geodesic_mesh
from .models_gnn import (DownConvLayers, GCNConvLayers, GNNConfig, GNNModel, UpConvLayers,
                         loss_func)

__all__ = [
    "GCNConv", "Linear", "GraphedForward", "KernelEvents", "StackForward", "event_bracket_overhead", "GraphCSR", "GraphCache", "default_cache", "prepare_graph", "Mesh",
    "complete_graph", "geodesic_mesh", "DownConvLayers", "GCNConvLayers", "GNNConfig", "GNNModel",
    "UpConvLayers", "loss_func", "InteractionNet", "InteractionForecaster", "EdgeGraph", "interaction_graph",
]
__version__ = "0.1.0"
