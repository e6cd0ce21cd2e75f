"""Input formats: the NC.inp parser against the reference's own parser tests (math-bem/src/core/io/nc_format.rs:
733-818: the SAMPLE_NC_INP project and its assertions), node / element files, and RoomConfig JSON against the reference's
example_rectangular.json (tests/golden/room_example_rectangular.json, a data file of the reference) with the mesh counts of
RectangularRoom::generate_mesh (geometry.rs:107-183)."""
import os
import numpy as np
from math_audio_amd import io as mio

HERE = os.path.dirname(os.path.abspath(__file__))

SAMPLE_NC_INP = """##-------------------------------------------
## This file was created by mesh2input
##-------------------------------------------
Mesh2HRTF 1.0.0
##
Test Description
##
## Controlparameter I
0 0 0 0 7 0
##
## Controlparameter II
1 1 0.000001 0.00e+00 1 0 0
##
## Load Frequency Curve
0 2
0.000000 0.000000e+00 0.0
0.000001 0.400000e+04 0.0
##
## 1. Main Parameters I
2 100 50 0 0 2 1 0 0
##
## 2. Main Parameters II
1 0 0 0.0000e+00 0 0 0
##
## 3. Main Parameters III
0 0 0 0
##
## 4. Main Parameters IV
343 1.21 1.0 0.0 0.0 0.0 0.0
##
NODES
nodes.txt
##
ELEMENTS
elements.txt
##
BOUNDARY
ELEM 0 TO 49 VELO 1.0 -1 0.0 -1
RETU
##
PLANE WAVES
1 0.0 -1.0 0.0 1.0 -1 0.0 -1
##
END
"""


def test_parse_nc_input_sample():                        # nc_format.rs:776-791, 809-817
    cfg = mio.parse_nc_input_string(SAMPLE_NC_INP, ".")
    assert "Mesh2HRTF" in cfg.version
    assert cfg.main_params_i["num_nodes"] == 100 and cfg.main_params_i["num_elements"] == 50 and cfg.main_params_i["solver_method"] == 1
    assert abs(cfg.main_params_iv["speed_of_sound"] - 343.0) < 0.01 and abs(cfg.main_params_iv["density"] - 1.21) < 0.01
    assert len(cfg.node_files) == 1 and len(cfg.element_files) == 1
    assert len(cfg.boundary_conditions) == 1 and len(cfg.plane_waves) == 1
    assert cfg.control_params_i == [0, 0, 0, 0, 7, 0] and cfg.control_params_ii[2] == 1e-6
    assert cfg.frequency_curve == [(0.0, 0.0, 0.0), (1e-6, 4000.0, 0.0)] and cfg.frequencies() == [4000.0]
    assert cfg.main_params_ii["preconditioner"] == 1 and cfg.main_params_iii == [0, 0, 0, 0]
    assert abs(cfg.wave_number(1000.0) - 2.0 * np.pi * 1000.0 / 343.0) < 1e-15


def test_parse_boundary_and_source_lines():              # nc_format.rs:793-807
    bc = mio.parse_boundary_line("ELEM 0 TO 100 VELO 1.0 -1 0.0 -1")
    assert bc["elem_start"] == 0 and bc["elem_end"] == 100 and bc["bc_type"] == "VELO" and abs(bc["value_re"] - 1.0) < 1e-3
    assert mio.parse_boundary_line("ELEM 0 TO x VELO 1.0 -1 0.0 -1") is None and mio.parse_boundary_line("NODE 1") is None
    cfg = mio.parse_nc_input_string(SAMPLE_NC_INP.replace("PLANE WAVES\n1 0.0 -1.0 0.0 1.0 -1 0.0 -1", "POINT SOURCES\n1 0.1 0.2 0.3 2.0 -1 0.5 -1"), ".")
    assert cfg.plane_waves == [] and cfg.point_sources[0]["position"] == [0.1, 0.2, 0.3] and cfg.point_sources[0]["amplitude_im"] == 0.5
    pw = mio.parse_nc_input_string(SAMPLE_NC_INP, ".").plane_waves[0]
    assert abs(pw["direction"][1] + 1.0) < 1e-3 and abs(pw["amplitude_re"] - 1.0) < 1e-3


def test_nc_project_to_mesh(tmp_path):
    """A whole project on disk: node file with ids and a count line, element file mixing Tri3 and Quad4, boundary values."""
    (tmp_path / "nodes.txt").write_text("6\n1 0.0 0.0 0.1\n2 1.0 0.0 0.1\n3 1.0 1.0 0.1\n4 0.0 1.0 0.1\n5 2.0 0.0 0.1\n6 2.0 1.0 0.1\n")
    (tmp_path / "elements.txt").write_text("2\n1 0 1 2 3 0 0 0\n2 1 4 5 -1 0 0\n")
    (tmp_path / "NC.inp").write_text(SAMPLE_NC_INP.replace("ELEM 0 TO 49 VELO 1.0 -1 0.0 -1", "ELEM 0 TO 0 VELO 0.5 -1 0.25 -1\nELEM 1 TO 1 PRES 2.0 -1 0.0 -1"))
    cfg = mio.parse_nc_input(str(tmp_path / "NC.inp"))
    assert np.array_equal(mio.load_nc_elements(cfg.element_files[0]), [[0, 1, 2, 3], [1, 4, 5, -1]])
    # (Quad4 rows end at the first negative entry; here the reference's take_while keeps the trailing zeros of the first row:
    #  `1 0 1 2 3 0 0 0` -> 7 node ids, ElementType::Quad4 with the first four used by every consumer)
    m = cfg.to_mesh()
    assert m.n_elem == 2 and m.nodes.shape == (6, 3)
    assert list(m.bc_type) == [0, 1] and m.bc_values[0, 0] == 0.5 + 0.25j and m.bc_values[1, 0] == 2.0
    assert abs(m.area[0] - 1.0) < 1e-15 and abs(m.area[1] - 0.5) < 1e-15 and np.allclose(np.abs(m.normal[:, 2]), 1.0)
    assert np.array_equal(mio.load_nc_nodes(cfg.node_files[0])[4], [2.0, 0.0, 0.1])
    (tmp_path / "bare.txt").write_text("0.5 0.5 0.5\n1.5 0.5 0.5\n")          # `x y z` rows without ids or a count
    assert mio.load_nc_nodes(str(tmp_path / "bare.txt")).shape == (2, 3)


def test_room_config_json():
    cfg = mio.RoomConfig.from_file(os.path.join(HERE, "golden", "room_example_rectangular.json"))
    assert (cfg.width, cfg.depth, cfg.height) == (5.0, 4.0, 2.5) and cfg.method == "gmres" and cfg.mesh_resolution == 2
    assert cfg.sources[0]["position"] == (2.5, 0.5, 1.2) and cfg.listening_positions == [(2.5, 2.0, 1.2)]
    f = cfg.generate_frequencies()
    assert len(f) == 50 and abs(f[0] - 50.0) < 1e-12 and abs(f[-1] - 1000.0) < 1e-9 and abs(f[1] / f[0] - f[2] / f[1]) < 1e-12
    nodes, conn = cfg.generate_mesh()
    nx, ny, nz = 10, 8, 5
    assert conn.shape == (2 * (nx * ny + nx * nz + ny * nz), 4)
    assert nodes.shape[0] == 2 * ((nx + 1) * (ny + 1) + (nx + 1) * (nz + 1) + (ny + 1) * (nz + 1))     # faces do not share nodes
    assert nodes.min() == 0.0 and np.allclose(nodes.max(axis=0), [5.0, 4.0, 2.5])
    d = dict(cfg.raw); d["frequencies"] = dict(min_freq=100.0, max_freq=200.0, num_points=3, spacing="linear"); d.pop("solver")
    c2 = mio.RoomConfig(d)
    assert c2.generate_frequencies() == [100.0, 150.0, 200.0] and c2.method == "direct" and c2.mesh_resolution == 2
