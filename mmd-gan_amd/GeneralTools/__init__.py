"""API mirror of the reference's GeneralTools package (names and call signatures only; every
numeric op runs as a HIP kernel through mmdgan_hip)."""
