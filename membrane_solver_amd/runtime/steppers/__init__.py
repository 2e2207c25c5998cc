from .conjugate_gradient import ConjugateGradient
from .gradient_descent import GradientDescent

__all__ = ["ConjugateGradient", "GradientDescent"]
