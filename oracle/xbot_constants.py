"""XBot-L hot-path constants, restated from the reference's config classes.

TEST INFRASTRUCTURE (oracle).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this package; the product path never does.

Every value cites the reference line it restates (paths relative to /root/reference/humanoid).
"""
import math

# envs/custom/humanoid_config.py:38-48
FRAME_STACK = 15
C_FRAME_STACK = 3
NUM_SINGLE_OBS = 47
SINGLE_NUM_PRIV_OBS = 73
NUM_ACTIONS = 12
NUM_DOF = 12
NUM_BODIES = 13          # 12 revolute joints + base after collapse_fixed_joints (legged_robot_config.py:106)
EPISODE_LENGTH_S = 24.0

# body indices in the rigid-body tensors (depth-first URDF order, inferred)
BASE_BODY = 0
FEET_BODIES = (6, 12)    # *ankle_roll*  (humanoid_config.py:58)
KNEE_BODIES = (4, 10)    # *knee*        (humanoid_config.py:59)

# envs/custom/humanoid_config.py:128-131 + legged_robot.py:711
SIM_DT = 0.001
DECIMATION = 10
DT = DECIMATION * SIM_DT                       # 0.01 (python double, as in the reference)
MAX_EPISODE_LENGTH = math.ceil(EPISODE_LENGTH_S / DT)   # 2400, legged_robot.py:717-718
RESAMPLE_STEPS = int(8.0 / DT)                 # 800, legged_robot.py:309 + humanoid_config.py:163
PUSH_INTERVAL = math.ceil(4 / DT)              # 400, legged_robot.py:720 + humanoid_config.py:150

# control, humanoid_config.py:118-126
ACTION_SCALE = 0.25
# joint order: L{roll,yaw,pitch,knee,ankle_pitch,ankle_roll}, R{same} (urdf/XBot-L.urdf:1415-2516)
P_GAINS = [200.0, 200.0, 350.0, 350.0, 15.0, 15.0] * 2
D_GAINS = [10.0] * 12
EFFORT = [100.0, 100.0, 250.0, 250.0, 100.0, 100.0] * 2
TORQUE_LIMIT_FACTOR = 0.85                     # humanoid_config.py:55, legged_robot.py:293
DOF_LOWER = [-0.44, -1.05, -1.57, -1.05, -0.70, -0.44, -1.57, -1.05, -1.31, -1.10, -0.87, -0.44]
DOF_UPPER = [1.57, 1.05, 1.31, 1.10, 0.87, 0.44, 0.44, 1.05, 1.57, 1.05, 0.70, 0.44]
DEFAULT_DOF_POS = [0.0] * 12                   # humanoid_config.py:101-115
BASE_INIT_STATE = [0.0, 0.0, 0.95, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]  # :98 + legged_robot_config.py:88-91
ENV_SPACING = 3.0                              # legged_robot_config.py:40

# normalisation, humanoid_config.py:218-227
OBS_SCALE_LIN_VEL = 2.0
OBS_SCALE_ANG_VEL = 1.0
OBS_SCALE_DOF_POS = 1.0
OBS_SCALE_DOF_VEL = 0.05
OBS_SCALE_QUAT = 1.0
CLIP_OBS = 18.0
CLIP_ACTIONS = 18.0

# noise, humanoid_config.py:84-95
NOISE_LEVEL = 0.6
NOISE_DOF_POS = 0.05
NOISE_DOF_VEL = 0.5
NOISE_ANG_VEL = 0.1
NOISE_QUAT = 0.03

# domain randomisation, humanoid_config.py:144-156
MAX_PUSH_VEL_XY = 0.2
MAX_PUSH_ANG_VEL = 0.4
ACTION_DELAY = 0.5
ACTION_NOISE = 0.02
FRICTION_RANGE = (0.1, 2.0)
ADDED_MASS_RANGE = (-5.0, 5.0)

# commands, humanoid_config.py:158-172
CMD_LIN_VEL_X = (-0.3, 0.6)
CMD_LIN_VEL_Y = (-0.3, 0.3)
CMD_HEADING = (-3.14, 3.14)
CMD_ANG_VEL_YAW = (-0.3, 0.3)      # humanoid_config.py ranges.ang_vel_yaw (used only with heading_command = False)

# rewards, humanoid_config.py:174-216
BASE_HEIGHT_TARGET = 0.89
MIN_DIST = 0.2
MAX_DIST = 0.5
TARGET_JOINT_POS_SCALE = 0.17
TARGET_FEET_HEIGHT = 0.06
CYCLE_TIME = 0.64
TRACKING_SIGMA = 5
MAX_CONTACT_FORCE = 700

# alphabetical = evaluation order, because class_to_dict iterates dir() (utils/helpers.py:44-59);
# every scale is multiplied by dt once (legged_robot.py:523-528), in python double arithmetic.
REWARD_SCALES_RAW = [
    ("action_smoothness", -0.002),
    ("base_acc", 0.2),
    ("base_height", 0.2),
    ("collision", -1.0),
    ("default_joint_pos", 0.5),
    ("dof_acc", -1e-7),
    ("dof_vel", -5e-4),
    ("feet_air_time", 1.0),
    ("feet_clearance", 1.0),
    ("feet_contact_forces", -0.01),
    ("feet_contact_number", 1.2),
    ("feet_distance", 0.2),
    ("foot_slip", -0.05),
    ("joint_pos", 1.6),
    ("knee_distance", 0.2),
    ("low_speed", 0.2),
    ("orientation", 1.0),
    ("torques", -1e-5),
    ("track_vel_hard", 0.5),
    ("tracking_ang_vel", 1.1),
    ("tracking_lin_vel", 1.2),
    ("vel_mismatch_exp", 0.5),
]
REWARD_NAMES = [n for n, _ in REWARD_SCALES_RAW]
REWARD_SCALES_DT = [s * DT for _, s in REWARD_SCALES_RAW]
NUM_REWARDS = len(REWARD_NAMES)

# PPO / policy, humanoid_config.py:230-251 + legged_robot_config.py:200-221
ACTOR_HIDDEN = [512, 256, 128]
CRITIC_HIDDEN = [768, 256, 128]
INIT_NOISE_STD = 1.0
GAMMA = 0.994
LAM = 0.9
CLIP_PARAM = 0.2
VALUE_LOSS_COEF = 1.0
ENTROPY_COEF = 0.001
LEARNING_RATE = 1e-5
MAX_GRAD_NORM = 1.0
DESIRED_KL = 0.01
NUM_LEARNING_EPOCHS = 2
NUM_MINI_BATCHES = 4
NUM_STEPS_PER_ENV = 60
SEED = 5
