from setuptools import setup, find_packages

# same packaging contract as the reference's feature/setup.py: module `diffusion_feature`, sub-package `components`
setup(name="diffusion_feature", version="0.1", packages=find_packages(), py_modules=["diffusion_feature"],
      package_data={"": ["libgdf.so", "configs/*.json"]})
