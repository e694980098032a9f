"""eabnet_amd -- MI355X (gfx950) implementation of EaBNet's per-frame causal
beamforming hot path behind the reference's Python call surface.

    from eabnet_amd import EaBNet, prepare_data, numParams, com_mag_mse_loss

Importing the package does not touch the GPU; the HIP library
(eabnet_amd/lib/libeabnet_hip.so) is loaded on first use and its absence is an
error, never a fallback.
"""
from .spec import GagConfig, NetConfig, gag_param_specs, param_specs  # noqa: F401
from .model import (EaBNet, GaGNet, EaBNetWithPostNet, make_gag_net, make_eabnet_with_postnet,  # noqa: F401
                    StreamingEnhancer, Pipeline, prepare_data, stft_compress, istft, filter_and_sum, numParams, com_mag_mse_loss,
                    stagewise_com_mag_mse_loss, eabnet_with_postnet_loss)

__all__ = ["EaBNet", "GaGNet", "EaBNetWithPostNet", "make_gag_net", "make_eabnet_with_postnet", "StreamingEnhancer", "Pipeline", "prepare_data",
           "stft_compress", "istft", "filter_and_sum", "numParams", "com_mag_mse_loss", "stagewise_com_mag_mse_loss",
           "eabnet_with_postnet_loss",
           "NetConfig", "GagConfig", "param_specs", "gag_param_specs"]
