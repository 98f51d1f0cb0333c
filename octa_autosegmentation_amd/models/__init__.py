"""Mirror of the reference's models/ package for the hot path: DynUNet + registry, losses, training step."""
