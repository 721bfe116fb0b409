from .conditions import CameraPreprocessor, euler_xyz_to_matrix, preprocess_smpl_fix_body  # noqa: F401
