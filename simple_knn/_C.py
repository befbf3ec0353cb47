"""``from simple_knn._C import distCUDA2`` (Garment_3DGS/gaussiansplatting/scene/gaussian_model.py:20) served by
the MI355X HIP kernel ``gd_scene_dist2`` -- same signature: [P,3] float CUDA tensor -> [P] mean squared distance
to the 3 nearest neighbours (simple-knn/spatial.cu:14-25)."""
from garmentdreamer_amd.gaussian_model import distCUDA2  # noqa: F401
