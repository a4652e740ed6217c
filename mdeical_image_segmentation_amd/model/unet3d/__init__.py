"""Mirror of the reference's `model.unet3d` package (an empty __init__ in the reference; sub-modules are imported by path)."""
