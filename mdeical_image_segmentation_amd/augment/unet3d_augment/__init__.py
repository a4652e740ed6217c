"""Mirror of the reference's `augment.unet3d_augment` package (on-device implementations)."""
