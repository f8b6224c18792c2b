from ppsurf_amd.modules import (FKAConvLayer, ResidualBlock, FKAConvNetwork, AttentionPoco, STN, PointNetfeat, MLP,  # noqa: F401
                                batch_gather, max_pool, interpolate, count_parameters)
