"""cracks_amd — MI355X-native Newton residual/Jacobian assembly for the phase-field
fracture system of tjhei/cracks (``assemble_system`` / ``assemble_nl_residual``,
cracks.cc:2129-2512).  See DESIGN.md."""

__version__ = "0.1.0"
