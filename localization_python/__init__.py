"""Top-level alias: `import localization_python` finds this build's drop-in for the reference's Python package
(localization_python/localization_python/__init__.py) -- LocalizationNode, main, and the submodules
localization_node and optimize_global_map_pose under the reference's own module paths."""
import sys

from slam_sensor_fusion_amd.localization_python import LocalizationNode, main, localization_node, optimize_global_map_pose  # noqa: F401

sys.modules[__name__ + ".localization_node"] = localization_node
sys.modules[__name__ + ".optimize_global_map_pose"] = optimize_global_map_pose
