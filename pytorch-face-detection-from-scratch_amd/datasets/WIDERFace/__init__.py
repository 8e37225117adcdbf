from .dataset import convert_bbx_to_feature_map, encode_batch  # noqa: F401
