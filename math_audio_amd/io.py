"""Input formats either side of the hot path: NumCalc / Mesh2HRTF `NC.inp` projects and RoomConfig JSON.

Host plumbing that turns the reference's file formats into the flat arrays of `ma_mesh_t` / the `ma_room_*` calls:
  parse_nc_input_string, parse_nc_input, load_nc_nodes, load_nc_elements   math-bem/src/core/io/nc_format.rs:204-694
  NcInput.to_mesh (elements + BOUNDARY specs -> MeshArrays)                the loop a driver writes around those functions
  RoomConfig (from_file / from_dict), rectangular_room_mesh                math-xem-common/src/config.rs:12-36, 583-604;
                                                                           geometry.rs:107-183, 434-469 (RectangularRoom)
Nothing here is timed; the parsers follow the reference's control flow line by line so that the same files load the same way.
"""
import json
import math
import os
import numpy as np

from . import MeshArrays
from . import mesh as _mesh


# ------------------------------------------------------------------ NC.inp (nc_format.rs)
def _ints(line):
    out = []
    for s in line.split():
        try:
            out.append(int(s))
        except ValueError:
            pass                                       # parse_int_line :522-527: tokens that do not parse are dropped
    return out


def _floats(line):
    out = []
    for s in line.split():
        try:
            out.append(float(s))
        except ValueError:
            pass                                       # parse_float_line :529-538
    return out


class NcInput:
    """NcInputConfig (nc_format.rs:20-56)."""

    def __init__(self, base_dir):
        self.version = ""; self.description = ""
        self.control_params_i = []; self.control_params_ii = []
        self.frequency_curve = []
        self.main_params_i = dict(element_type=0, num_nodes=0, num_elements=0, num_object_files=0, num_eval_files=0, bc_type=0, solver_method=0, fmm_method=0, parallel=0)
        self.main_params_ii = dict(preconditioner=0, iterative_solver=0, reserved1=0, reserved2=0.0, output_level=0, reserved3=0, reserved4=0)
        self.main_params_iii = []
        self.main_params_iv = dict(speed_of_sound=343.0, density=1.21, reference_pressure=1.0, reserved=[])
        self.node_files = []; self.element_files = []
        self.symmetry = None
        self.boundary_conditions = []; self.plane_waves = []; self.point_sources = []
        self.base_dir = base_dir

    def frequencies(self):
        """The frequency steps of the curve: entries with a positive frequency (0.000001 -> 4000 Hz in the sample)."""
        return [f for (_, f, _) in self.frequency_curve if f > 0.0]

    def wave_number(self, frequency):                  # to_physics_params :698-705 -> PhysicsParams::new
        return 2.0 * math.pi * frequency / self.main_params_iv["speed_of_sound"]

    def to_mesh(self):
        """Nodes and elements of every listed file, geometry from compute_element_geometry (generators.rs:513-602), boundary
        values from the BOUNDARY specs (ELEM a TO b VELO / PRES re curve im curve; other kinds leave the element rigid)."""
        nodes = np.concatenate([load_nc_nodes(p) for p in self.node_files], axis=0) if self.node_files else np.zeros((0, 3))
        conn = np.concatenate([load_nc_elements(p) for p in self.element_files], axis=0) if self.element_files else np.zeros((0, 4), dtype=np.int32)
        center, normal, area = _mesh.element_geometry(nodes, conn)
        n = conn.shape[0]
        bc_type = np.zeros(n, dtype=np.uint8); bc_values = np.zeros((n, 4), dtype=np.complex128); bc_len = np.ones(n, dtype=np.int32)
        for bc in self.boundary_conditions:
            lo, hi = max(0, bc["elem_start"]), min(n - 1, bc["elem_end"])
            if bc["bc_type"] == "VELO":
                bc_type[lo:hi + 1] = 0
            elif bc["bc_type"] == "PRES":
                bc_type[lo:hi + 1] = 1
            else:
                continue
            bc_values[lo:hi + 1, 0] = complex(bc["value_re"], bc["value_im"])
        return MeshArrays(nodes, conn, center, normal, area, bc_type=bc_type, bc_values=bc_values, bc_len=bc_len)


def parse_nc_input_string(content, base_dir="."):
    """parse_nc_input_string (nc_format.rs:213-520)."""
    cfg = NcInput(base_dir)
    lines = content.splitlines()
    i = 0

    def prev_has(idx, key, *excluded):
        if idx <= 0:
            return False
        p = lines[idx - 1]
        return key in p and not any(e in p for e in excluded)
    while i < len(lines):
        line = lines[i].strip()
        if not line or line.startswith("#"):
            i += 1
            continue
        if line.startswith("Mesh2HRTF"):
            cfg.version = line
            i += 1
            continue
        if not cfg.version:
            i += 1
            continue
        if prev_has(i, "Controlparameter I", "Controlparameter II"):
            cfg.control_params_i = _ints(line); i += 1; continue
        if prev_has(i, "Controlparameter II"):
            cfg.control_params_ii = _floats(line); i += 1; continue
        if prev_has(i, "Frequency Curve"):
            header = _ints(line)
            npts = header[1] if len(header) > 1 else 0
            i += 1
            for _ in range(npts):
                if i < len(lines):
                    v = _floats(lines[i])
                    if len(v) >= 3:
                        cfg.frequency_curve.append((v[0], v[1], v[2]))
                    i += 1
            continue
        if prev_has(i, "Main Parameters I", "Main Parameters II", "Main Parameters III", "Main Parameters IV"):
            v = _ints(line)
            keys = ("element_type", "num_nodes", "num_elements", "num_object_files", "num_eval_files", "bc_type", "solver_method", "fmm_method", "parallel")
            cfg.main_params_i = {k: (v[q] if q < len(v) else 0) for q, k in enumerate(keys)}
            i += 1; continue
        if prev_has(i, "Main Parameters II", "Main Parameters III", "Main Parameters IV"):
            v = _floats(line)
            g = lambda q, d=0.0: v[q] if q < len(v) else d
            cfg.main_params_ii = dict(preconditioner=int(g(0)), iterative_solver=int(g(1)), reserved1=int(g(2)), reserved2=g(3), output_level=int(g(4)), reserved3=int(g(5)), reserved4=int(g(6)))
            i += 1; continue
        if prev_has(i, "Main Parameters III", "Main Parameters IV"):
            cfg.main_params_iii = _ints(line); i += 1; continue
        if prev_has(i, "Main Parameters IV"):
            v = _floats(line)
            cfg.main_params_iv = dict(speed_of_sound=v[0] if len(v) > 0 else 343.0, density=v[1] if len(v) > 1 else 1.21,
                                      reference_pressure=v[2] if len(v) > 2 else 1.0, reserved=v[3:])
            i += 1; continue
        if line in ("NODES", "ELEMENTS"):
            target = cfg.node_files if line == "NODES" else cfg.element_files
            i += 1
            while i < len(lines):
                ln = lines[i].strip()
                if ln.startswith("##") or not ln:
                    break
                if not ln.startswith("#"):
                    target.append(os.path.join(cfg.base_dir, ln))
                i += 1
            continue
        if line == "SYMMETRY":
            i += 1
            if i < len(lines) and not lines[i].strip().startswith("#"):
                flags = _ints(lines[i].strip()); i += 1
                if i < len(lines):
                    org = _floats(lines[i].strip())
                    cfg.symmetry = dict(flags=[(flags[q] if q < len(flags) else 0) != 0 for q in range(3)], origin=[(org[q] if q < len(org) else 0.0) for q in range(3)])
                    i += 1
            continue
        if line == "BOUNDARY":
            i += 1
            while i < len(lines):
                ln = lines[i].strip()
                if ln.startswith("##") or ln == "RETU":
                    i += 1
                    break
                if ln.startswith("#") or not ln:
                    i += 1
                    continue
                bc = parse_boundary_line(ln)
                if bc is not None:
                    cfg.boundary_conditions.append(bc)
                i += 1
            continue
        if line in ("PLANE WAVES", "POINT SOURCES"):
            plane = line == "PLANE WAVES"
            i += 1
            while i < len(lines):
                ln = lines[i].strip()
                if ln.startswith("##") or not ln:
                    break
                if not ln.startswith("#"):
                    v = _floats(ln)
                    if len(v) >= 8:                    # parse_plane_wave_line :573-587 / parse_point_source_line :589-603
                        rec = dict(amplitude_re=v[4], curve_re=int(v[5]), amplitude_im=v[6], curve_im=int(v[7]))
                        rec["direction" if plane else "position"] = [v[1], v[2], v[3]]
                        (cfg.plane_waves if plane else cfg.point_sources).append(rec)
                i += 1
            continue
        if line == "END":
            break
        i += 1
    return cfg


def parse_boundary_line(line):
    """ELEM start TO end TYPE value curve value curve (nc_format.rs:545-571)."""
    p = line.split()
    if len(p) >= 9 and p[0] == "ELEM" and p[2] == "TO":
        try:
            return dict(elem_start=int(p[1]), elem_end=int(p[3]), bc_type=p[4], value_re=float(p[5]), curve_re=int(p[6]), value_im=float(p[7]), curve_im=int(p[8]))
        except ValueError:
            return None
    return None


def parse_nc_input(path):
    with open(path) as f:
        return parse_nc_input_string(f.read(), os.path.dirname(os.path.abspath(path)))


def _data_lines(path):
    with open(path) as f:
        lines = f.read().splitlines()
    if not lines:
        return []
    try:
        int(lines[0].strip())                          # an optional leading count line (:616-621, :651-655)
        return lines[1:]
    except ValueError:
        return lines


def load_nc_nodes(path):
    """load_nc_nodes (nc_format.rs:605-635): `id x y z` or `x y z` per line."""
    out = []
    for ln in _data_lines(path):
        v = _floats(ln)
        if len(v) >= 4:
            out.append(v[1:4])
        elif len(v) >= 3:
            out.append(v[0:3])
    return np.array(out, dtype=np.float64).reshape(-1, 3)


def load_nc_elements(path):
    """load_nc_elements (nc_format.rs:638-694): `id n1 n2 n3 [n4]`, node lists end at the first negative entry; rows of 4 ints,
    -1 in the fourth slot of a triangle."""
    out = []
    for ln in _data_lines(path):
        v = _ints(ln)
        if len(v) >= 4:
            c = []
            for t in v[1:]:
                if t < 0:
                    break
                c.append(t)
            c = c[:4]
            out.append(c + [-1] * (4 - len(c)))
    return np.array(out, dtype=np.int32).reshape(-1, 4)


# ------------------------------------------------------------------ RoomConfig JSON (math-xem-common/src/config.rs)
class RoomConfig:
    """RoomConfig::from_file (config.rs:583-594) with serde's defaults: solver.method "direct", mesh_resolution 2, spacing
    "logarithmic", source amplitude 1, rigid boundaries."""

    def __init__(self, d):
        self.raw = d
        room = d["room"]
        self.room_type = room["type"]
        if self.room_type == "rectangular":
            self.width, self.depth, self.height = float(room["width"]), float(room["depth"]), float(room["height"])
        elif self.room_type == "lshaped":
            self.lshaped = {k: float(room[k]) for k in ("width1", "depth1", "width2", "depth2", "height")}
        else:
            raise ValueError("unknown room type %r" % self.room_type)
        self.sources = [dict(name=s["name"], position=(float(s["position"]["x"]), float(s["position"]["y"]), float(s["position"]["z"])),
                             amplitude=float(s.get("amplitude", 1.0))) for s in d["sources"]]
        self.listening_positions = [(float(p["x"]), float(p["y"]), float(p["z"])) for p in d["listening_positions"]]
        fr = d["frequencies"]
        self.min_freq, self.max_freq, self.num_points = float(fr["min_freq"]), float(fr["max_freq"]), int(fr["num_points"])
        self.spacing = fr.get("spacing", "logarithmic")
        sv = d.get("solver", {})
        self.method = sv.get("method", "direct"); self.mesh_resolution = int(sv.get("mesh_resolution", 2))
        self.adaptive_integration = bool(sv.get("adaptive_integration", False))
        g = sv.get("gmres", {})
        self.gmres = dict(max_iter=int(g.get("max_iter", 100)), restart=int(g.get("restart", 50)), tolerance=float(g.get("tolerance", 1e-6)))

    @staticmethod
    def from_file(path):
        with open(path) as f:
            return RoomConfig(json.load(f))

    def generate_frequencies(self):                    # FrequencyConfig::generate_frequencies (config.rs:358-366)
        if self.spacing.lower() == "linear":
            if self.num_points < 2:
                return [self.min_freq]
            return [self.min_freq + (self.max_freq - self.min_freq) * float(i) / float(self.num_points - 1) for i in range(self.num_points)]
        return _mesh.log_space(self.min_freq, self.max_freq, self.num_points)

    def generate_mesh(self):
        if self.room_type != "rectangular":
            raise NotImplementedError("L-shaped rooms: only the rectangular generator is restated")
        return rectangular_room_mesh(self.width, self.depth, self.height, self.mesh_resolution)


def rectangular_room_mesh(width, depth, height, elements_per_meter):
    """RectangularRoom::generate_mesh (geometry.rs:107-183) with add_surface_mesh (:434-469): six gridded faces, quads,
    nodes NOT shared between faces. Returns (nodes [n, 3], conn [m, 4]) for ma_room_element_data / ma_room_build_matrix_adaptive."""
    nx = int(math.ceil(width * float(elements_per_meter))); ny = int(math.ceil(depth * float(elements_per_meter))); nz = int(math.ceil(height * float(elements_per_meter)))
    nodes, conn = [], []

    def surface(o, u, v, nu, nv):
        base = len(nodes)
        for j in range(nv + 1):
            for i in range(nu + 1):
                a = float(i) / float(nu); b = float(j) / float(nv)
                nodes.append([o[0] + a * (u[0] - o[0]) + b * (v[0] - o[0]), o[1] + a * (u[1] - o[1]) + b * (v[1] - o[1]), o[2] + a * (u[2] - o[2]) + b * (v[2] - o[2])])
        for j in range(nv):
            for i in range(nu):
                n0 = base + j * (nu + 1) + i
                conn.append([n0, n0 + 1, base + (j + 1) * (nu + 1) + i + 1, base + (j + 1) * (nu + 1) + i])
    w, d, h = width, depth, height
    surface((0.0, 0.0, 0.0), (w, 0.0, 0.0), (0.0, d, 0.0), nx, ny)        # floor
    surface((0.0, 0.0, h), (w, 0.0, h), (0.0, d, h), nx, ny)              # ceiling
    surface((0.0, 0.0, 0.0), (w, 0.0, 0.0), (0.0, 0.0, h), nx, nz)        # front wall
    surface((0.0, d, 0.0), (w, d, 0.0), (0.0, d, h), nx, nz)              # back wall
    surface((0.0, 0.0, 0.0), (0.0, d, 0.0), (0.0, 0.0, h), ny, nz)        # left wall
    surface((w, 0.0, 0.0), (w, d, 0.0), (w, 0.0, h), ny, nz)              # right wall
    return np.array(nodes, dtype=np.float64), np.array(conn, dtype=np.int32)
