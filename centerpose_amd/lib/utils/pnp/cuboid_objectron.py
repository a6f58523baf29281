"""Cuboid model of the reference (utils/pnp/cuboid_objectron.py:11-110): vertex order and frame."""
from enum import IntEnum


class CuboidVertexType(IntEnum):
    FrontTopRight = 0
    FrontTopLeft = 1
    FrontBottomLeft = 2
    FrontBottomRight = 3
    RearTopRight = 4
    RearTopLeft = 5
    RearBottomLeft = 6
    RearBottomRight = 7
    Center = 8
    TotalCornerVertexCount = 8
    TotalVertexCount = 9


class Cuboid3d(object):
    """Axis-aligned box centred at the origin: x = width (right +), y = height (top +), z = depth (front +)."""

    def __init__(self, size3d=(1.0, 1.0, 1.0), coord_system=None, parent_object=None):
        self.center_location = [0, 0, 0]
        self.coord_system = coord_system
        self.size3d = size3d
        self._vertices = [0, 0, 0] * 8
        self.generate_vertexes()

    def get_vertex(self, vertex_type):
        return self._vertices[vertex_type]

    def get_vertices(self):
        return self._vertices

    def generate_vertexes(self):
        width, height, depth = self.size3d
        cx, cy, cz = self.center_location
        right, left = cx + width / 2.0, cx - width / 2.0
        top, bottom = cy + height / 2.0, cy - height / 2.0
        front, rear = cz + depth / 2.0, cz - depth / 2.0
        self._vertices = [[left, bottom, rear], [left, bottom, front], [left, top, rear], [left, top, front],
                          [right, bottom, rear], [right, bottom, front], [right, top, rear], [right, top, front]]
