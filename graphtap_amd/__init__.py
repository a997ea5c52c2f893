"""graphtap_amd -- MI355X-native SpMV vertex-program engine behind GraphTap's Vertex_Program surface.

The compute path is libgraphtap_amd.so (hand-written HIP for gfx950, C ABI in include/graphtap_amd.h).
This package is the host-side mirror of the reference's operator interface (Graph, Vertex_Program and
the five apps' programs). Importing it never touches the GPU; using it without the built library or
without a GPU raises GraphTapError -- there is no CPU fallback."""
from ._lib import GT_INF as INF, GraphOptions, GraphTapError, ProgramOptions  # noqa: F401
from .graph import _2D_, _2DT_, _CSC_, _DCSC_, _TCSC_, _TCSC_CF_, Graph  # noqa: F401
from .vertex_program import (_COL_, _ROW_, BFS_Program, CC_Program, Deg_Program, PR_Program, SSSP_Program,  # noqa: F401
                             Vertex_Program)
