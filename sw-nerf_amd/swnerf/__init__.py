"""swnerf - MI355X-native NeRF volumetric renderer behind the Python call surface of
daihangpku/SW-NeRF (get_rays / ndc_rays / sample_pdf / raw2outputs / Embedder /
NeRF MLPs / run_network / render_rays / render).  See DESIGN.md and INTEGRATION.md."""
from . import synth  # noqa: F401  (numpy only)

__all__ = ["synth", "ray", "embedder", "model", "render"]
