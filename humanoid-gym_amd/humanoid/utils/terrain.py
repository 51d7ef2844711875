"""Procedural terrain is outside the MI355X hot path (XBot-L trains on `plane`, and the
height field only feeds PhysX collision).  The names exist so `from humanoid.utils import Terrain` keeps working."""


class Terrain:
    def __init__(self, cfg=None, num_robots=0):
        raise NotImplementedError("height-field / trimesh terrain is not part of the MI355X hot path; use mesh_type='plane'")


class HumanoidTerrain(Terrain):
    pass
