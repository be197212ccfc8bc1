"""API mirror of the reference's DeepLearning package."""
