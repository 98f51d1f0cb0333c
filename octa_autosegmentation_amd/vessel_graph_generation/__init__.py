"""Mirror of the reference's vessel_graph_generation package (hot-path parts only)."""
