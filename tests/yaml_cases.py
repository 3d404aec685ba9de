"""Settings-file variants for the DepthModule parser: the golden KITTI file with keys replaced / removed, and the
outcomes (b_parse_LiDAR, b_parse_LiDARUpsampling) the reference's own parser produces for them
(tests/test_reference_build.py runs the reference source; tests/test_shim.py runs the drop-in class)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_YAML = os.path.join(ROOT, "tests", "golden", "KITTI00-02.yaml")


def yaml_with(tmp_path, name, edits=None, drop=()):
    edits = dict(edits or {})
    out = []
    for line in open(GOLDEN_YAML).read().splitlines():
        key = line.split(":")[0].strip()
        if key in drop:
            continue
        if key in edits:
            line = "%s: %s" % (key, edits.pop(key))
        out.append(line)
    out += ["%s: %s" % kv for kv in edits.items()]
    path = os.path.join(str(tmp_path), name)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return path


# (edits, dropped keys, (LiDAR parse ok, up-sampling parse ok or None when not reached with a defined method))
PARSE_CASES = [
    ({}, (), (True, True)),
    ({}, ("LiDAR.Tr23",), (False, None)),                                   # missing key -> LiDAR parse fails
    ({}, ("Camera.bf",), (False, None)),
    ({"LiDAR.max_dist": "200"}, (), (False, None)),                         # an int node is "not a real number"
    ({"LiDAR.Method": '"IPBasic"'}, (), (True, False)),                     # declared, never implemented upstream
    ({"LiDAR.Method": '"Bogus"'}, (), (False, None)),
    ({}, ("LiDAR.Method",), (False, None)),
    ({"LiDAR.Method": '"None"'}, (), (True, True)),
    ({"LiDAR.Method": '"NearestNeighborPixel"'}, ("LiDAR.MethodNearestNeighborPixel.SearchDistance",), (True, False)),
    ({"LiDAR.Method": '"AverageFiltering"'}, ("LiDAR.MethodInverseDilation.KernelType",), (True, True)),   # string read never throws
    ({"LiDAR.Method": '"AverageFiltering"'}, ("LiDAR.MethodInverseDilation.KernelSize_u",), (True, False)),  # case falls through
    ({"LiDAR.Method": '"AverageFiltering"', "LiDAR.MethodAverageFiltering.bDoDilationPreprocessing": "2"}, (), (True, False)),
    ({"LiDAR.Method": '"AverageFiltering"', "LiDAR.MethodAverageFiltering.bDoDilationPreprocessing": "0"},
     ("LiDAR.MethodAverageFiltering.DilationPreprocessing_KernelSize",), (True, True)),
    ({"LiDAR.Method": '"AverageFiltering"'}, ("LiDAR.MethodAverageFiltering.DilationPreprocessing_KernelSize",), (True, False)),
    ({"LiDAR.MethodInverseDilation.KernelSize_v": "7"}, (), (True, False)),
]
