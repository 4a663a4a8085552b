from .scalekd import ScaleKD

__all__ = ["ScaleKD"]
