"""
Triangulated-mesh optics: host-side set-up (mesh generation, triangulation, lookup tables,
Clough-Tocher data) with NumPy/SciPy exactly where the reference uses them; the per-ray work
(Moller-Trumbore over the coarse faces, nearest fine point, <= 8 candidate faces, C1 cubic
interpolation) runs on the device from the flattened tables (xicsrt_amd/scene.py).

  ShapeMesh            generic mesh from user arrays      xicsrt/optics/_ShapeMesh.py:21-262
  ShapeMeshSphere      x-y grid on a sphere               xicsrt/optics/_ShapeMeshSphere.py:17-98
  ShapeMeshCylinder    (x, angle) grid on a cylinder      xicsrt/optics/_ShapeMeshCylinder.py:19-192
  ShapeMeshTorus       (major, minor) angle grid          xicsrt/optics/_ShapeMeshTorus.py:18-268
"""
import numpy as np

from . import ShapeObject


def _vector_rotate(a, b, theta):
    """Rotate vector a about vector b by theta (xicsrt/tools/xicsrt_math.py:72-112, 1-D case)."""
    a = np.asarray(a)
    b = np.asarray(b)
    b_hat = b / np.linalg.norm(b)
    u = b_hat * np.dot(a, b_hat)
    v = a - u
    w = np.cross(b_hat, v)
    return u + v * np.cos(theta) + w * np.sin(theta)


def mesh_tables(points, normals, faces, interpolate):
    """
    Everything the device needs about one mesh (reference: ShapeMesh._mesh_precalc,
    _ShapeMesh.py:198-261): faces (given or the x-y Delaunay), per-face vertices / edges /
    normal / area term, the point -> faces table (<= 8, ascending), and for interpolation
    the x-y Delaunay with SciPy's Clough-Tocher vertex gradients for z and the normals.
    """
    from scipy.spatial import Delaunay
    from scipy.interpolate import CloughTocher2DInterpolator

    points = np.ascontiguousarray(points, dtype=np.float64)
    out = {'points': points}
    delaunay = Delaunay(points[:, 0:2])
    if faces is None:
        faces = delaunay.simplices
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    out['faces'] = faces
    if interpolate:
        nrm = np.asarray(normals, dtype=np.float64)
        interps = [CloughTocher2DInterpolator(delaunay, points[:, 2].flatten())]
        interps += [CloughTocher2DInterpolator(delaunay, nrm[:, k].flatten()) for k in range(3)]
        out['ct_simplices'] = np.ascontiguousarray(delaunay.simplices, dtype=np.int32)
        out['ct_neighbors'] = np.ascontiguousarray(delaunay.neighbors, dtype=np.int32)
        out['ct_transform'] = np.ascontiguousarray(delaunay.transform, dtype=np.float64)       # [T][3][2]
        out['ct_points'] = np.ascontiguousarray(delaunay.points, dtype=np.float64)             # [P][2]
        out['ct_values'] = np.ascontiguousarray([np.asarray(i.values)[:, 0] for i in interps], dtype=np.float64)
        out['ct_grad'] = np.ascontiguousarray([np.asarray(i.grad)[:, 0, :] for i in interps], dtype=np.float64)
        out['ct_vertex_simplex'] = np.ascontiguousarray(delaunay.vertex_to_simplex, dtype=np.int32)
    p0 = points[faces[..., 0], :]
    p1 = points[faces[..., 1], :]
    p2 = points[faces[..., 2], :]
    faces_normal = np.cross((p0 - p1), (p2 - p1))
    faces_normal /= np.linalg.norm(faces_normal, axis=1)[:, None]
    out['p0'] = np.ascontiguousarray(p0)
    out['p1'] = np.ascontiguousarray(p1)
    out['p2'] = np.ascontiguousarray(p2)
    out['edge1'] = np.ascontiguousarray(p1 - p0)
    out['edge2'] = np.ascontiguousarray(p2 - p0)
    out['faces_normal'] = np.ascontiguousarray(faces_normal)
    out['faces_area'] = np.ascontiguousarray(np.linalg.norm(np.cross((p0 - p1), (p0 - p2)), axis=1))
    n = len(points)
    p_faces_idx = np.zeros((8, n), dtype=np.int32)
    p_faces_mask = np.zeros((8, n), dtype=np.uint8)
    order = np.argsort(faces.ravel(), kind='stable')
    pts_sorted = faces.ravel()[order]
    rows = (order // 3).astype(np.int32)
    starts = np.searchsorted(pts_sorted, np.arange(n), side='left')
    ends = np.searchsorted(pts_sorted, np.arange(n), side='right')
    for ip in range(n):
        ii_f = np.sort(rows[starts[ip]:ends[ip]])
        if len(ii_f) > 8:
            raise ValueError('mesh point %d belongs to more than 8 faces' % ip)
        p_faces_idx[:len(ii_f), ip] = ii_f
        p_faces_mask[:len(ii_f), ip] = 1
    out['p_faces_idx'] = p_faces_idx
    out['p_faces_mask'] = p_faces_mask
    return out


class ShapeMesh(ShapeObject):
    """A surface given as a triangulated point cloud (optionally with normals and a coarse pre-selection mesh)."""

    shape_kind = 'mesh'

    def default_config(self):
        """
        mesh_points, mesh_normals, mesh_faces                 : the mesh (faces default to the x-y Delaunay)
        mesh_coarse_points, mesh_coarse_normals, mesh_coarse_faces : coarse mesh for pre-selection
        mesh_interpolate : C1 interpolation of z and normals (default: when normals are given)
        mesh_refine      : two-level search (default: when a coarse mesh is given)
        """
        config = super().default_config()
        for key in ('mesh_points', 'mesh_normals', 'mesh_faces', 'mesh_coarse_points', 'mesh_coarse_normals',
                    'mesh_coarse_faces', 'mesh_interpolate', 'mesh_refine'):
            config[key] = None
        return config

    def check_param(self):
        super().check_param()
        p = self.param
        if p['mesh_interpolate'] is None:
            p['mesh_interpolate'] = (p['mesh_normals'] is not None)
        elif p['mesh_interpolate']:
            if p['mesh_normals'] is None:
                raise Exception('Surface normal vectors must be defined in order to use mesh interpolation.')
        if p['mesh_refine'] is None:
            if p['mesh_coarse_points'] is not None:
                p['mesh_refine'] = True
        pts = np.asarray(p['mesh_points'])
        spread = [np.max(pts[:, k]) - np.min(pts[:, k]) for k in range(3)]
        if (spread[2] > spread[0]) or (spread[2] > spread[1]):
            self.log.warning('Mesh is not oriented with the surface normals near the local z direction.\n'
                             'This may lead to unexpected and incorrect results.')

    def initialize(self):
        super().initialize()
        p = self.param
        p['mesh'] = mesh_tables(p['mesh_points'], p['mesh_normals'], p['mesh_faces'], p['mesh_interpolate'])
        if p['mesh_coarse_points'] is not None:
            p['mesh_coarse'] = mesh_tables(p['mesh_coarse_points'], p['mesh_coarse_normals'],
                                           p['mesh_coarse_faces'], p['mesh_interpolate'])


class _GridMesh(ShapeMesh):
    """Meshes generated on a 2-parameter grid; faces = Delaunay triangulation of the parameter grid."""

    def surface(self, a, b):
        raise NotImplementedError

    def grid_ranges(self):
        raise NotImplementedError

    def generate_mesh(self, mesh_size):
        from scipy.spatial import Delaunay
        a_range, b_range = self.grid_ranges()
        num_a, num_b = int(mesh_size[0]), int(mesh_size[1])
        a = np.linspace(a_range[0], a_range[1], num_a)
        b = np.linspace(b_range[0], b_range[1], num_b)
        aa, bb = np.meshgrid(a, b, indexing='ij')
        xyz = np.empty((num_a, num_b, 3))
        nrm = np.empty((num_a, num_b, 3))
        for ia in range(num_a):
            for ib in range(num_b):
                xyz[ia, ib], nrm[ia, ib] = self.surface(aa[ia, ib], bb[ia, ib])
        angles_2d = np.stack((aa.flatten(), bb.flatten()), axis=0).T
        points = np.stack((xyz[..., 0].flatten(), xyz[..., 1].flatten(), xyz[..., 2].flatten())).T
        normals = np.stack((nrm[..., 0].flatten(), nrm[..., 1].flatten(), nrm[..., 2].flatten())).T
        faces = Delaunay(angles_2d).simplices
        return points, normals, faces

    def _generate_both(self):
        p = self.param
        p['mesh_points'], p['mesh_normals'], p['mesh_faces'] = self.generate_mesh(p['mesh_size'])
        (p['mesh_coarse_points'], p['mesh_coarse_normals'],
         p['mesh_coarse_faces']) = self.generate_mesh(p['mesh_coarse_size'])


class ShapeMeshSphere(ShapeMesh):

    def default_config(self):
        """radius, mesh_size, mesh_coarse_size; built in local coordinates (trace_local = True)."""
        config = super().default_config()
        config['radius'] = 1.0
        config['mesh_size'] = (11, 11)
        config['mesh_coarse_size'] = (5, 5)
        config['trace_local'] = True
        return config

    def setup(self):
        super().setup()
        p = self.param
        p['mesh_points'], p['mesh_normals'], p['mesh_faces'] = self.generate_mesh(p['mesh_size'])
        (p['mesh_coarse_points'], p['mesh_coarse_normals'],
         p['mesh_coarse_faces']) = self.generate_mesh(p['mesh_coarse_size'])

    def generate_mesh(self, meshsize):
        """Grid over the (x, y) footprint, lifted onto the sphere that touches the local origin from above.
        (The products below are fed to the device bit for bit, so the expressions keep the operand order of
        xicsrt/optics/_ShapeMeshSphere.py:60-98: r - sqrt(r^2 - x^2 - y^2), normal = (centre - point) / length.)"""
        from scipy.spatial import Delaunay
        radius = self.param['radius']
        half = (self.param['xsize'] / 2, self.param['ysize'] / 2)
        gx, gy = np.meshgrid(np.linspace(-half[0], half[0], int(meshsize[0])),
                             np.linspace(-half[1], half[1], int(meshsize[1])))
        sag = radius - np.sqrt(radius ** 2 - gx ** 2 - gy ** 2)
        points = np.stack((gx.flatten(), gy.flatten(), sag.flatten())).T
        inward = np.array([0.0, 0.0, radius]) - points
        inward /= np.expand_dims(np.linalg.norm(inward, axis=1), 1)
        return points, inward, Delaunay(points[:, 0:2]).simplices


class ShapeMeshCylinder(_GridMesh):

    def default_config(self):
        """radius, mesh_size, mesh_coarse_size, mesh_xsize, mesh_ysize; local coordinates."""
        config = super().default_config()
        config['mesh_refine'] = True
        config['mesh_size'] = (11, 11)
        config['mesh_coarse_size'] = (5, 5)
        config['mesh_xsize'] = None
        config['mesh_ysize'] = None
        config['radius'] = 1.0
        config['trace_local'] = True
        return config

    def setup(self):
        super().setup()
        p = self.param
        xsize = p['xsize'] if p['mesh_xsize'] is None else p['mesh_xsize']
        ysize = p['ysize'] if p['mesh_ysize'] is None else p['mesh_ysize']
        p['x_range'] = [-1 * xsize / 2, xsize / 2]
        half_angle = np.arcsin(ysize / 2 / (p['radius']))
        p['angle_range'] = [-1 * half_angle, half_angle]
        self._generate_both()

    def grid_ranges(self):
        return self.param['x_range'], self.param['angle_range']

    def surface(self, x, angle):
        """Point and inward normal at axial position x and angle about the cylinder axis (local x, through
        (0, 0, radius)).  Operand order as in xicsrt/optics/_ShapeMeshCylinder.py:150-192: the axis point is
        0 + radius * z + (x, 0, 0) and the surface point is that minus radius * normal."""
        radius = self.param['radius']
        up, along = np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0])
        on_axis = np.array([0.0, 0.0, 0.0]) + radius * up + np.array([x, 0.0, 0.0])
        normal = _vector_rotate(up, along, angle)
        return on_axis - radius * normal, normal


class ShapeMeshTorus(_GridMesh):

    def default_config(self):
        """radius_major, radius_minor, convex, normal_method, mesh_size, mesh_coarse_size, mesh_xsize, mesh_ysize."""
        config = super().default_config()
        config['mesh_refine'] = True
        config['mesh_size'] = (11, 11)
        config['mesh_coarse_size'] = (5, 5)
        config['mesh_xsize'] = None
        config['mesh_ysize'] = None
        config['radius_major'] = 1.0
        config['radius_minor'] = 0.2
        config['convex'] = [False, False]
        config['normal_method'] = 'analytic'
        config['trace_local'] = True
        return config

    def setup(self):
        super().setup()
        p = self.param
        signs = {(False, False): (1, 1), (False, True): (1, -1), (True, False): (-1, 1), (True, True): (-1, -1)}
        try:
            key = tuple(bool(v) for v in np.asarray(p['convex']).tolist())
            p['torus_sign_major'], p['torus_sign_minor'] = signs[key]
        except Exception:
            raise Exception(f"Cannot be parse convex config option: {p['convex']}")
        xsize = p['xsize'] if p['mesh_xsize'] is None else p['mesh_xsize']
        ysize = p['ysize'] if p['mesh_ysize'] is None else p['mesh_ysize']
        half_major = np.arcsin(xsize / 2 / (p['radius_major']))
        half_minor = np.arcsin(ysize / 2 / p['radius_minor'])
        p['angle_major'] = [-1 * half_major, half_major]
        p['angle_minor'] = [-1 * half_minor, half_minor]
        if p['normal_method'] != 'analytic':
            raise NotImplementedError("normal_method '%s' is not implemented" % p['normal_method'])
        self._generate_both()

    def grid_ranges(self):
        return self.param['angle_major'], self.param['angle_minor']

    def surface(self, a, b):
        """Point and normal at major angle a (about local y) and minor angle b (about the tube's tangent).
        Two rotations: `up` about y by a gives the direction from the torus centre line to the tube centre,
        that direction about the tube's tangent by b gives the surface normal.  The sign pair (major, minor)
        selects which of the four torus patches faces the origin.  Operand order as in
        xicsrt/optics/_ShapeMeshTorus.py:132-160 (the generated points are compared bit for bit)."""
        p = self.param
        sign_major, sign_minor = p['torus_sign_major'], p['torus_sign_minor']
        big, small = p['radius_major'], p['radius_minor']
        up = np.asarray([0.0, 0.0, 1.0])
        hinge = np.cross(up, np.asarray([1.0, 0.0, 0.0]))              # local y: axis of the major circle
        spoke = _vector_rotate(up, hinge, a)
        tube_centre = big * up * sign_major - big * spoke * sign_major + small * spoke * sign_minor
        tangent = np.cross(spoke * sign_minor, hinge)
        normal = _vector_rotate(spoke * sign_minor, tangent, b)
        return tube_centre - normal * small, normal
