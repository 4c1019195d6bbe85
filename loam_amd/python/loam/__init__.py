"""`import loam` — the Python surface of the reference (python/loam_bindings.cpp) on the MI355X back end."""
from .loam_python import *  # noqa: F401,F403
from .loam_python import (FeatureExtractionParams, LidarParams, LoamFeatures, Pose3d, Quaterniond,  # noqa: F401
                          RegistrationDetail, RegistrationIterationInfo, RegistrationParams,
                          RegistrationTerminationType, computeCurvature, computeValidPoints, extractFeatures,
                          registerFeatures)
