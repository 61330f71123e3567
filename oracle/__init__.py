"""TEST INFRASTRUCTURE: CPU oracle of the PPO hot path (see ppo_oracle.py header). Never imported by the product."""
