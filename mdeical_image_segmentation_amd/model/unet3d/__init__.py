"""Mirror of the reference's `model.unet3d` package (empty __init__ in the reference)."""
