"""Register this package under the module names the reference imports, so the reference's
model files and test_animating scripts run unchanged on MI355X:

    from models import softsplat                                             (animating_softmax_splating.py:26)
    from models.projection.euler_integration_manipulator import EulerIntegration, euler_integration   (:9)
"""
import sys
import types


def install_into_reference():
    """Call once before importing the reference's model modules."""
    from . import euler_integration_manipulator, softsplat
    sys.modules["models.softsplat"] = softsplat
    sys.modules["models.projection.euler_integration_manipulator"] = euler_integration_manipulator
    models = sys.modules.get("models")
    if models is not None:
        models.softsplat = softsplat
    proj = sys.modules.get("models.projection")
    if proj is not None:
        proj.euler_integration_manipulator = euler_integration_manipulator
    # the reference does `import cupy` only inside models/softsplat.py; nothing else needs it
    return softsplat, euler_integration_manipulator
