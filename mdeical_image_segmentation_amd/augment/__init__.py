"""Mirror of the reference's `augment` package."""
