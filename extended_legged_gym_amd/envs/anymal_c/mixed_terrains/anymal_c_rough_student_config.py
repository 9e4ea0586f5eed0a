"""`anymal_c_rough_student` task config (values of the reference's
`envs/anymal_c/mixed_terrains/anymal_c_rough_student_config.py:5-71`): the student sees a history of proprioceptive rows, the
teacher the current row plus the height scan."""
from extended_legged_gym_amd.envs.base.legged_robot_config import LeggedRobotCfgPPO
from .anymal_c_rough_config import AnymalCRoughCfg


class AnymalCRoughStudentCfg(AnymalCRoughCfg):
    class env(AnymalCRoughCfg.env):
        num_observations = 144           # 48 (proprio) x 3 (history)
        num_privileged_obs = 235         # 48 (proprio) + 187 (height scan)
        history_length = 3

    class terrain(AnymalCRoughCfg.terrain):
        mesh_type = 'trimesh'
        measure_heights = True
        measured_points_x = [-0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
        measured_points_y = [-0.5, -0.4, -0.3, -0.2, -0.1, 0., 0.1, 0.2, 0.3, 0.4, 0.5]
        horizontal_scale = 0.1
        vertical_scale = 0.005
        border_size = 25
        curriculum = True
        static_friction = 1.0
        dynamic_friction = 1.0
        restitution = 0.


class AnymalCRoughStudentCfgPPO(LeggedRobotCfgPPO):
    seed = 1
    runner_class_name = 'OnPolicyRunner'

    class policy(LeggedRobotCfgPPO.policy):
        init_noise_std = 1.0
        teacher_hidden_dims = [512, 256, 128]
        student_hidden_dims = [512, 256, 128]
        activation = 'elu'

    class algorithm:
        num_learning_epochs = 1
        gradient_length = 15
        learning_rate = 1e-3
        max_grad_norm = 1.0
        loss_type = "mse"

    class runner(LeggedRobotCfgPPO.runner):
        policy_class_name = 'StudentTeacher'
        algorithm_class_name = 'Distillation'
        num_steps_per_env = 24
        max_iterations = 1500
        teacher_model_path = ""          # path of a trained teacher checkpoint (the reference hard-codes a developer's home directory)
        save_interval = 50
        experiment_name = 'rough_anymal_c_student'
        run_name = ''
        resume = False
        load_run = -1
        checkpoint = -1
        resume_path = None
