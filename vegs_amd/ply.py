"""Gaussian-model PLY files in the layout the reference writes and reads (scene/gaussian_model.py:182-259):
one `vertex` element of float32 properties  x y z nx ny nz f_dc_0..2 f_rest_0..(3*(K-1)-1) opacity scale_0..2
rot_0..3, binary little endian.  f_dc / f_rest are stored channel-major (the reference's
`transpose(1, 2).flatten(start_dim=1)`), i.e. f_rest_j = features_rest[:, j % (K-1), j // (K-1)].

Host-side I/O only (numpy): a file written here loads with the reference's load_ply and vice versa.  The
reference goes through `plyfile` and a per-row Python tuple conversion (minutes at 2 M Gaussians); this is one
contiguous [N, 62] float32 write.
"""
import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    """construct_list_of_attributes (scene/gaussian_model.py:182-194)"""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
            + ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def save_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Arguments as the reference's raw parameters: xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3],
    opacity [N,1] (logit), scaling [N,3] (log), rotation [N,4].  numpy arrays or tensors."""
    a = [np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, np.float32)
         for t in (xyz, features_dc, features_rest, opacity, scaling, rotation)]
    xyz, f_dc, f_rest, opacity, scaling, rotation = a
    n = xyz.shape[0]
    if f_dc.shape != (n, 1, 3) or f_rest.ndim != 3 or f_rest.shape[0] != n or f_rest.shape[2] != 3:
        raise ValueError("features_dc must be [N,1,3] and features_rest [N,K-1,3]")
    if opacity.shape != (n, 1) or scaling.shape[0] != n or rotation.shape[0] != n:
        raise ValueError("opacity must be [N,1]; scaling / rotation need N rows")
    f_dc = f_dc.transpose(0, 2, 1).reshape(n, -1)
    f_rest = f_rest.transpose(0, 2, 1).reshape(n, -1)
    rows = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opacity, scaling, rotation), axis=1).astype("<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    assert rows.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {k}\n" for k in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        rows.tofile(f)


def read_vertex_table(path):
    """{property name: 1-D array} of the first element of a binary-little-endian PLY file."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported in the vertex element")
                props.append((tok[2], "<" + _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: only binary_little_endian PLY is supported (got {fmt})")
        data = np.fromfile(f, dtype=np.dtype(props), count=count)
        if data.shape[0] != count:
            raise ValueError(f"{path}: truncated vertex data")
    return {name: data[name] for name, _ in props}


def load_ply(path, max_sh_degree=3):
    """-> dict of float32 arrays shaped like the reference's parameters after load_ply
    (scene/gaussian_model.py:218-259): xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3], opacity [N,1],
    scaling [N,3], rotation [N,4]."""
    t = read_vertex_table(path)
    by_index = lambda prefix: sorted((k for k in t if k.startswith(prefix)), key=lambda k: int(k.split("_")[-1]))
    xyz = np.stack([t["x"], t["y"], t["z"]], 1)
    n = xyz.shape[0]
    f_dc = np.stack([t["f_dc_0"], t["f_dc_1"], t["f_dc_2"]], 1).reshape(n, 3, 1)
    rest_names = by_index("f_rest_")
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties, expected {3 * (max_sh_degree + 1) ** 2 - 3} "
                         f"for max_sh_degree {max_sh_degree}")
    f_rest = np.stack([t[k] for k in rest_names], 1).reshape(n, 3, (max_sh_degree + 1) ** 2 - 1) if rest_names \
        else np.zeros((n, 3, 0), np.float32)
    out = {"xyz": xyz, "features_dc": f_dc.transpose(0, 2, 1), "features_rest": f_rest.transpose(0, 2, 1),
           "opacity": np.asarray(t["opacity"])[:, None], "scaling": np.stack([t[k] for k in by_index("scale_")], 1),
           "rotation": np.stack([t[k] for k in by_index("rot")], 1)}
    return {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}
