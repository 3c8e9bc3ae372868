"""Mirror of the reference's ``sdf`` package interface (reference sdf/sdf/sdf.py:6-26): ``SDF().forward(faces,
vertices, grid_size)`` -> phi[B, G, G, G], evaluated by libmvfit (mvfit_sdf).  Like the reference binding the
number of triangles is ``faces.size(0)`` and there is no backward (the op is called under ``no_grad``,
code/utils/fitting.py:361; the reference's backward returns None, sdf.py:17-19)."""
from __future__ import annotations

import torch

from .engine import MvFit, MvFitError


class SDF(torch.nn.Module):
    def __init__(self, engine: MvFit):
        super().__init__()
        self._eng = engine

    @torch.no_grad()
    def forward(self, faces, vertices, grid_size=32):
        if not isinstance(vertices, torch.Tensor) or not vertices.is_cuda:
            raise RuntimeError('vertices must be a CUDA tensor')            # sdf_cuda.cpp:3 (CHECK_CUDA)
        if not vertices.is_contiguous() or (isinstance(faces, torch.Tensor) and not faces.is_contiguous()):
            raise RuntimeError('inputs must be contiguous')                  # sdf_cuda.cpp:4 (CHECK_CONTIGUOUS)
        return self._eng.sdf(faces, vertices, grid_size)


def sdf(engine: MvFit, faces, vertices, grid_size=32):
    return SDF(engine)(faces, vertices, grid_size)


__all__ = ['SDF', 'sdf', 'MvFitError']
