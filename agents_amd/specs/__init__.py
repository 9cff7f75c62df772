"""Spec types (see tensor_spec.py)."""
from agents_amd.specs.tensor_spec import (ArraySpec, BoundedArraySpec, BoundedTensorSpec,
                                          TensorSpec)
from agents_amd.specs import tensor_spec
array_spec = tensor_spec  # the reference has two modules; both map to the same classes here
