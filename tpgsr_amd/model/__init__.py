from . import tsrn, stn_head, tps_spatial_transformer  # noqa: F401
