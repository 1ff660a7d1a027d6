"""Spherical convolution operator package: ``sphere_conv`` (module / autograd Function, Python side of the operator seam) and
``sphere_conv_cuda`` (the two-function native seam, backed by libmode_hip.so)."""
from . import sphere_conv_cuda
from .sphere_conv import SphereConv, SphereConvFunction

__all__ = ['SphereConv', 'SphereConvFunction', 'sphere_conv_cuda']
