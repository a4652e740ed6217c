"""Mirror of the helpers of model/unet3d/utils.py that the hot path needs (:109-110)."""


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]
