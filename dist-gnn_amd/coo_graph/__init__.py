from .parted_coo_graph import COO_Graph, Parted_COO_Graph, GraphCache, BasicGraph

__all__ = ['COO_Graph', 'Parted_COO_Graph', 'GraphCache', 'BasicGraph']
