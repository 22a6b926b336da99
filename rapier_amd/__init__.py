"""rapier_amd — MI355X-native `PhysicsPipeline::step()` hot path for rapier3d scenes."""
from . import scenes  # noqa: F401
from .world import (  # noqa: F401
    ColliderHandle, ColliderSet, ImpulseJointSet, IntegrationParameters, PhysicsPipeline, PhysicsWorld,
    RapierHipError, RigidBodyHandle, RigidBodySet, ShardComm, step_many,
)
