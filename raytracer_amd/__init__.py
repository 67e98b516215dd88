"""MI355X-native ray-tracing pass: a drop-in for the render pass of monkeysoop/raytracer behind a C ABI.

`api` binds librt_hotpath.so (include/rt_hotpath.h); `camera` / `params` produce the per-frame uniforms; `scenes`
provides scene inputs; `dist` shards frames over GPUs.  The library is loaded lazily and there is no CPU fallback.
"""
from . import api, camera, params  # noqa: F401

__all__ = ["api", "camera", "params"]
